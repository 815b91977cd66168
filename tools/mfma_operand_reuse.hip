// Measurement aid: does the ORDER in which a GEMM's MFMAs visit their operand fragments change what the power-limited
// matrix pipes sustain?  A bare v_mfma_f32_32x32x16_f16 stream (one wave per SIMD, 16 accumulators, 8 A and 8 B random
// fragments in registers) issued in different visiting orders:
//   0 constant operands                 1 both operands change at every MFMA (amp_calibrate_mfma_f16's random mode)
//   2 A held for 8 MFMAs, B changes     3 B held for 8 MFMAs, A changes
//   4 A changes every MFMA, B every 2nd (the 4x2 wave tile's a-outer/b-inner loop with A = weights)
//   5 snake: exactly one operand changes per MFMA            6 = 1 with half of B's values zero (post-ReLU activations)
//   7 = 1 with the low 5 mantissa bits of both operands zero
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_operand_reuse.hip -o tools/bin/mfma_operand_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float fx16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void stream_kernel(float* out, int iters, unsigned long long* clocks) {
  fx16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  h8 a[8], b[8];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) {
        a[k][i] = (_Float16)0.5f;
        b[k][i] = (_Float16)0.25f;
      } else {
        s = s * 1664525u + 1013904223u;
        _Float16 va = (_Float16)(((int)((s >> 9) & 0xFFFF) - 32768) * (1.0f / 32768.0f));
        s = s * 1664525u + 1013904223u;
        _Float16 vb = (_Float16)(((int)((s >> 9) & 0xFFFF) - 32768) * (1.0f / 32768.0f));
        if (MODE == 6 && ((s >> 20) & 1)) vb = (_Float16)0.0f;
        if (MODE == 7) {
          unsigned short ua = __builtin_bit_cast(unsigned short, va) & 0xFFE0, ub = __builtin_bit_cast(unsigned short, vb) & 0xFFE0;
          va = __builtin_bit_cast(_Float16, ua);
          vb = __builtin_bit_cast(_Float16, ub);
        }
        a[k][i] = va;
        b[k][i] = vb;
      }
    }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      int ia, ib;
      if (MODE == 2) { ia = u >> 3; ib = u & 7; }
      else if (MODE == 3) { ia = u & 7; ib = u >> 3; }
      else if (MODE == 4) { ia = u & 7; ib = (u >> 1) & 7; }
      else if (MODE == 5) { ia = ((u + 1) >> 1) & 7; ib = (u >> 1) & 7; }
      else { ia = u & 7; ib = (u * 3 + (u >> 3)) & 7; }
      acc[u & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc[u & 15], 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = w1 - w0; }
  float t = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE>
static void run(int cus, int iters, float* out, unsigned long long* clocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ms;
  for (int rep = 0; rep < 7; ++rep) {
    hipEventRecord(e0, 0);
    stream_kernel<MODE><<<cus, 256>>>(out, iters, clocks);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
  }
  std::vector<unsigned long long> h(2 * cus);
  hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost);
  double mhz = 0;
  for (int i = 0; i < cus; ++i) mhz += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
  mhz /= cus;
  std::sort(ms.begin(), ms.end());
  const double flops = (double)cus * 4.0 * iters * 64.0 * 32768.0;
  printf("mode %d  median %.4f ms = %.0f TFLOP/s   best %.0f   core clock (last rep) %.0f MHz\n", MODE, ms[3], flops / (ms[3] * 1e-3) / 1e12,
         flops / (ms[0] * 1e-3) / 1e12, mhz);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 512;
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; unsigned long long* clocks;
  hipMalloc(&out, sizeof(float) * 256 * cus);
  hipMalloc(&clocks, sizeof(unsigned long long) * 2 * cus);
  for (int pass = 0; pass < 2; ++pass) {
    printf("pass %d (iters %d, %d CUs)\n", pass, iters, cus);
    run<0>(cus, iters, out, clocks); run<1>(cus, iters, out, clocks); run<2>(cus, iters, out, clocks); run<3>(cus, iters, out, clocks);
    run<4>(cus, iters, out, clocks); run<5>(cus, iters, out, clocks); run<6>(cus, iters, out, clocks); run<7>(cus, iters, out, clocks);
  }
  return 0;
}
