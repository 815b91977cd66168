// Measurement aid: does the MFMA SHAPE change what the power-limited matrix pipes sustain?  Same bare stream as
// tools/mfma_operand_reuse.hip (one wave per SIMD, random fp16 operands changing at every MFMA), once with
// v_mfma_f32_32x32x16_f16 (16 accumulators of 16 registers) and once with v_mfma_f32_16x16x32_f16 (32 accumulators of 4):
// same MACs per instruction-cycle, half the accumulator traffic per MAC, twice the operand traffic.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_power.hip -o tools/bin/mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float fx16 __attribute__((ext_vector_type(16)));
typedef float fx4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int RANDOM, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void stream_kernel(float* out, int iters, unsigned long long* clocks) {
  constexpr int NA32 = THREADS == 256 ? 16 : 8, NA16 = THREADS == 256 ? 32 : 16;  // accumulators per wave (2 waves / SIMD: half the registers)
  h8 a[8], b[8];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s = s * 1664525u + 1013904223u;
      a[k][i] = RANDOM ? (_Float16)(((int)((s >> 9) & 0xFFFF) - 32768) * (1.0f / 32768.0f)) : (_Float16)0.5f;
      s = s * 1664525u + 1013904223u;
      b[k][i] = RANDOM ? (_Float16)(((int)((s >> 9) & 0xFFFF) - 32768) * (1.0f / 32768.0f)) : (_Float16)0.25f;
    }
  float t = 0;
  unsigned long long c0, c1, w0, w1;
  if (SHAPE == 0) {
    fx16 acc[NA32];
#pragma unroll
    for (int i = 0; i < NA32; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    c0 = __builtin_readcyclecounter(); w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u)
        acc[u & (NA32 - 1)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 7], b[(u * 3 + (u >> 3)) & 7], acc[u & (NA32 - 1)], 0, 0, 0);
    }
    c1 = __builtin_readcyclecounter(); w1 = wall_clock64();
#pragma unroll
    for (int i = 0; i < NA32; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[i][r];
  } else {
    fx4 acc[NA16];
#pragma unroll
    for (int i = 0; i < NA16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;
    c0 = __builtin_readcyclecounter(); w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 128; ++u)  // 128 x (16 x 16 x 32) = 64 x (32 x 32 x 16) MACs
        acc[u & (NA16 - 1)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 7], b[(u * 3 + (u >> 3)) & 7], acc[u & (NA16 - 1)], 0, 0, 0);
    }
    c1 = __builtin_readcyclecounter(); w1 = wall_clock64();
#pragma unroll
    for (int i = 0; i < NA16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) t += acc[i][r];
  }
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = w1 - w0; }
  out[blockIdx.x * THREADS + threadIdx.x] = t;
}

template <int SHAPE, int RANDOM, int THREADS = 256>
static void run(int cus, int iters, float* out, unsigned long long* clocks) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ms;
  for (int rep = 0; rep < 7; ++rep) {
    (void)hipEventRecord(e0, 0);
    stream_kernel<SHAPE, RANDOM, THREADS><<<cus, THREADS>>>(out, iters, clocks);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float t; (void)hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
  }
  std::vector<unsigned long long> h(2 * cus);
  (void)hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost);
  double mhz = 0;
  for (int i = 0; i < cus; ++i) mhz += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
  mhz /= cus;
  std::sort(ms.begin(), ms.end());
  const double flops = (double)cus * (THREADS / 64) * iters * 64.0 * 32768.0;
  printf("%d waves/SIMD %s %s operands: median %.4f ms = %.0f TFLOP/s, core clock %.0f MHz\n", THREADS / 256, SHAPE ? "16x16x32" : "32x32x16",
         RANDOM ? "random  " : "constant", ms[3], flops / (ms[3] * 1e-3) / 1e12, mhz);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 512;
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; unsigned long long* clocks;
  (void)hipMalloc(&out, sizeof(float) * 512 * cus);
  (void)hipMalloc(&clocks, sizeof(unsigned long long) * 2 * cus);
  for (int pass = 0; pass < 2; ++pass) {
    run<0, 0>(cus, iters, out, clocks); run<1, 0>(cus, iters, out, clocks);
    run<0, 1>(cus, iters, out, clocks); run<1, 1>(cus, iters, out, clocks);
    run<0, 0, 512>(cus, iters, out, clocks); run<1, 0, 512>(cus, iters, out, clocks);
    run<0, 1, 512>(cus, iters, out, clocks); run<1, 1, 512>(cus, iters, out, clocks);
  }
  return 0;
}
