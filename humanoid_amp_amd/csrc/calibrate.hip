// Measurement aid (no reference counterpart; like the kernel tracer): what the matrix pipes of THIS device sustain on a bare
// stream of v_mfma_f32_32x32x16_f16 -- the instruction the discriminator GEMMs issue -- with one wave per SIMD and sixteen
// independent accumulators, on constant operands and on operands that change from one MFMA to the next (eight rotating
// register sets of full-entropy random fp16 values: what a GEMM on real data feeds the multipliers).  MI355X is
// power-limited: the second figure is 0.55-0.62 of the nominal 2 516.8 TFLOP/s, the first 0.70-0.84
// (profiles/r02_mfma_entropy_ceiling.txt), so a GEMM kernel's fraction of the NOMINAL peak has to be read against it.
// bench.py reports both next to roofline.frac.
#include "amp_common.hpp"

typedef _Float16 cal_h8 __attribute__((ext_vector_type(8)));
typedef float cal_fx16 __attribute__((ext_vector_type(16)));

namespace amp {

template <int ENTROPY>
__global__ __launch_bounds__(256, 1) void mfma_f16_calibration_kernel(float* out, int iters, unsigned long long* clocks) {
  cal_fx16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  cal_h8 a[8], b[8];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (ENTROPY) {
        s = s * 1664525u + 1013904223u;
        a[k][i] = (_Float16)(((int)((s >> 9) & 0xFFFF) - 32768) * (1.0f / 32768.0f));
        s = s * 1664525u + 1013904223u;
        b[k][i] = (_Float16)(((int)((s >> 9) & 0xFFFF) - 32768) * (1.0f / 32768.0f));
      } else {
        a[k][i] = (_Float16)(0.5f);
        b[k][i] = (_Float16)(0.25f);
      }
    }
  // shader clock (s_memtime: counts at the CURRENT core clock) against the constant 100 MHz wall clock (s_memrealtime):
  // their ratio over the loop is the clock the chip actually sustained under this load
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48; ++u)
      acc[u & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 7], b[(u * 3 + (u >> 3)) & 7], acc[u & 15], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0 && clocks) {
    clocks[2 * blockIdx.x] = c1 - c0;
    clocks[2 * blockIdx.x + 1] = w1 - w0;
  }
  float t = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

// The layer-2 stream since round 3: v_mfma_f32_16x16x32_f16, TWO waves per SIMD (512 threads), sixteen accumulators of four
// registers per wave -- the shape and occupancy of the discriminator's dominant kernels.  Same MACs per instruction-cycle as the
// 32 x 32 x 16 stream; on operands that change from one MFMA to the next it sustains ~13 % more (1.82-1.84 vs 1.62 PFLOP/s).
typedef float cal_fx4 __attribute__((ext_vector_type(4)));
template <int ENTROPY>
__global__ __launch_bounds__(512, 1) void mfma_f16_calibration16_kernel(float* out, int iters, unsigned long long* clocks) {
  cal_fx4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = cal_fx4{0.0f, 0.0f, 0.0f, 0.0f};
  cal_h8 a[8], b[8];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (ENTROPY) {
        s = s * 1664525u + 1013904223u;
        a[k][i] = (_Float16)(((int)((s >> 9) & 0xFFFF) - 32768) * (1.0f / 32768.0f));
        s = s * 1664525u + 1013904223u;
        b[k][i] = (_Float16)(((int)((s >> 9) & 0xFFFF) - 32768) * (1.0f / 32768.0f));
      } else {
        a[k][i] = (_Float16)(0.5f);
        b[k][i] = (_Float16)(0.25f);
      }
    }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 96; ++u)  // 96 x (16 x 16 x 32) = 48 x (32 x 32 x 16) MACs
      acc[u & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 7], b[(u * 3 + (u >> 3)) & 7], acc[u & 15], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0 && clocks) {
    clocks[2 * blockIdx.x] = c1 - c0;
    clocks[2 * blockIdx.x + 1] = w1 - w0;
  }
  float t = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) t += acc[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = t;
}

}  // namespace amp

extern "C" {

int amp_calibrate_mfma_f16(int32_t random_operands, int32_t iters, float* scratch_dev, int64_t scratch_floats, double* flops_out,
                           amp_stream_t stream) {
  AMP_REQUIRE(scratch_dev && flops_out, "amp_calibrate_mfma_f16: null argument");
  AMP_REQUIRE(iters >= 1 && iters <= (1 << 20), "amp_calibrate_mfma_f16: iters out of range");
  int dev = 0, cus = 0;
  AMP_HIP(hipGetDevice(&dev));
  AMP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  hipStream_t st = (hipStream_t)stream;
  if (random_operands & 2) {  // bit 1: the 16 x 16 x 32 / two-waves-per-SIMD stream (layer 2's shape); bit 0: random operands
    AMP_REQUIRE(scratch_floats >= (int64_t)cus * 512, "amp_calibrate_mfma_f16: scratch must hold %d floats", cus * 512);
    unsigned long long* clocks16 = scratch_floats >= (int64_t)cus * 512 + (int64_t)cus * 4
                                       ? reinterpret_cast<unsigned long long*>(scratch_dev + (int64_t)cus * 512) : nullptr;
    { amp::TraceScope trace__("mfma_f16_calibration_kernel", st);
      if (random_operands & 1) amp::mfma_f16_calibration16_kernel<1><<<(unsigned)cus, 512, 0, st>>>(scratch_dev, iters, clocks16);
      else amp::mfma_f16_calibration16_kernel<0><<<(unsigned)cus, 512, 0, st>>>(scratch_dev, iters, clocks16);
    }
    *flops_out = (double)cus * 8.0 * (double)iters * 96.0 * 16384.0;  // CUs x 8 waves x iters x 96 MFMAs x 2*16*16*32
    return amp::launch_status("mfma_f16_calibration_kernel");
  }
  AMP_REQUIRE(scratch_floats >= (int64_t)cus * 256, "amp_calibrate_mfma_f16: scratch must hold %d floats", cus * 256);
  // clock samples: 2 x uint64 per workgroup (shader-clock ticks, 100 MHz wall ticks) behind the 256 floats per CU
  unsigned long long* clocks = scratch_floats >= (int64_t)cus * 256 + (int64_t)cus * 4
                                   ? reinterpret_cast<unsigned long long*>(scratch_dev + (int64_t)cus * 256) : nullptr;
  { amp::TraceScope trace__("mfma_f16_calibration_kernel", st);
    if (random_operands) amp::mfma_f16_calibration_kernel<1><<<(unsigned)cus, 256, 0, st>>>(scratch_dev, iters, clocks);
    else amp::mfma_f16_calibration_kernel<0><<<(unsigned)cus, 256, 0, st>>>(scratch_dev, iters, clocks);
  }
  *flops_out = (double)cus * 4.0 * (double)iters * 48.0 * 32768.0;  // one launch: CUs x 4 waves x iters x 48 MFMAs x 2*32*32*16
  return amp::launch_status("mfma_f16_calibration_kernel");
}

}  // extern "C"
