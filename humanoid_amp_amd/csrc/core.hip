// libamp_engine.so: ABI version, error string, device probe.
#include "amp_common.hpp"

#include <cstring>

namespace amp {
char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace amp

extern "C" {

int amp_abi_version(void) { return AMP_ABI_VERSION; }

const char* amp_last_error(void) { return amp::last_error_buf(); }

int amp_device_name(char* buf, int64_t buf_len) {
  if (!buf || buf_len <= 0) return amp::fail(AMP_ERR_INVALID, "amp_device_name: null buffer");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return amp::fail(AMP_ERR_NO_DEVICE, "no HIP device visible");
  int dev = 0;
  AMP_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  AMP_HIP(hipGetDeviceProperties(&prop, dev));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return amp::fail(AMP_ERR_NO_DEVICE, "device %d is %s; libamp_engine.so is built for gfx950 only", dev, prop.gcnArchName);
  std::snprintf(buf, (size_t)buf_len, "%s (%s)", prop.name, prop.gcnArchName);
  return AMP_OK;
}

}  // extern "C"
