// libamp_engine.so: ABI version, error string, device probe.
#include "amp_common.hpp"

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace amp {
char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace amp

namespace amp {
namespace {
struct TraceRec {
  const char* name;
  hipEvent_t start, stop;
};
std::vector<TraceRec> g_recs;   // capacity fixed at amp_trace_begin: events are created there, never in a launch
int64_t g_used = 0;
bool g_on = false;
std::string g_filter;
std::mutex g_mu;  // launches may come from several host threads (one per stream): record slots are handed out under it
int64_t g_every = 1, g_seen = 0;  // sampling: only every g_every-th matching launch is bracketed (amp_trace_sample)
}  // namespace

bool trace_enabled() { return g_on; }

int trace_open(const char* kernel, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_on || g_used >= (int64_t)g_recs.size()) return -1;
  if (!g_filter.empty() && std::strstr(kernel, g_filter.c_str()) == nullptr) return -1;
  if (g_every > 1 && (g_seen++ % g_every) != 0) return -1;  // an event pair costs the queue a few us: sample
  const int i = (int)g_used++;
  g_recs[i].name = kernel;
  (void)hipEventRecord(g_recs[i].start, st);
  return i;
}

void trace_close(int rec, hipStream_t st) { (void)hipEventRecord(g_recs[rec].stop, st); }
}  // namespace amp

extern "C" {

int amp_trace_begin(int64_t capacity, const char* filter) {
  using namespace amp;
  if (capacity < 0 || capacity > (1 << 20)) return fail(AMP_ERR_INVALID, "amp_trace_begin: capacity out of range");
  std::lock_guard<std::mutex> lock(g_mu);
  g_on = false;
  for (auto& r : g_recs) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  g_recs.clear();
  g_used = 0;
  g_recs.resize((size_t)capacity);
  for (auto& r : g_recs) {
    r.name = "";
    // timing events only: no system-scope fence at the record (the default flag's cache write-back / invalidate delayed the
    // kernel behind a record by ~8 us in a rocprofv3 timeline: the tracer perturbed what it measured)
    if (hipEventCreateWithFlags(&r.start, hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&r.stop, hipEventDisableSystemFence) != hipSuccess)
      return fail(AMP_ERR_HIP, "amp_trace_begin: hipEventCreate failed");
  }
  g_filter = filter ? filter : "";
  g_every = 1;
  g_seen = 0;
  g_on = capacity > 0;
  return AMP_OK;
}

int amp_trace_sample(int64_t every) {
  using namespace amp;
  if (every < 1) return fail(AMP_ERR_INVALID, "amp_trace_sample: every must be >= 1");
  std::lock_guard<std::mutex> lock(g_mu);
  g_every = every;
  g_seen = 0;
  return AMP_OK;
}

int amp_trace_end(void) {
  std::lock_guard<std::mutex> lock(amp::g_mu);
  amp::g_on = false;
  return AMP_OK;
}

int64_t amp_trace_count(void) { return amp::g_used; }

int amp_trace_get(int64_t i, char* name_buf, int64_t name_len, float* ms) {
  using namespace amp;
  if (i < 0 || i >= g_used || !ms) return fail(AMP_ERR_INVALID, "amp_trace_get: bad index");
  hipError_t e = hipEventElapsedTime(ms, g_recs[i].start, g_recs[i].stop);
  if (e != hipSuccess) return fail(AMP_ERR_HIP, "amp_trace_get: %s", hipGetErrorString(e));
  if (name_buf && name_len > 0) std::snprintf(name_buf, (size_t)name_len, "%s", g_recs[i].name);
  return AMP_OK;
}

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
