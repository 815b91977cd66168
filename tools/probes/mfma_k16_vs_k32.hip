// Is v_mfma_f32_16x16x16_f16 on the low half of a k-block bit-identical to v_mfma_f32_16x16x32_f16 with the high half zero, and
// how long does each take?  (A candidate for the fused kernel's last layer-1 k-block: K D = 166 leaves 6 real columns in it.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_k16_vs_k32 tools/probes/mfma_k16_vs_k32.hip && tools/bin/mfma_k16_vs_k32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// A [16][32], B [16][32] (row = output row / column, k contiguous), C [16][16]; kmax = number of non-zero k (<= 16)
__global__ void compare_kernel(const _Float16* A, const _Float16* B, const float* C, float* D32, float* D16, int trials) {
  const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  for (int t = 0; t < trials; ++t) {
    const _Float16* a = A + (size_t)t * 512 + i * 32;
    const _Float16* b = B + (size_t)t * 512 + i * 32;
    h8 a8, b8;
    h4 a4, b4;
    for (int j = 0; j < 8; ++j) { a8[j] = a[8 * kq + j]; b8[j] = b[8 * kq + j]; }
    for (int j = 0; j < 4; ++j) { a4[j] = a[4 * kq + j]; b4[j] = b[4 * kq + j]; }
    f4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(size_t)t * 256 + (4 * kq + r) * 16 + i];
    const f4 d32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
    const f4 d16 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
      D32[(size_t)t * 256 + (4 * kq + r) * 16 + i] = d32[r];
      D16[(size_t)t * 256 + (4 * kq + r) * 16 + i] = d16[r];
    }
  }
}

template <int WIDE>
__global__ void timing_kernel(float* out, int iters) {
  h8 a8, b8;
  h4 a4, b4;
  for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(0.001f * (threadIdx.x + j)); b8[j] = (_Float16)(0.002f * (threadIdx.x ^ j)); }
  for (int j = 0; j < 4; ++j) { a4[j] = a8[j]; b4[j] = b8[j]; }
  f4 c[8];
  for (int k = 0; k < 8; ++k) c[k] = f4{0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (WIDE) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[k], 0, 0, 0);
      else c[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[k], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(iters * 8);
}

int main() {
  const int trials = 4096;
  for (int kmax : {6, 16}) {
    std::vector<_Float16> A((size_t)trials * 512), B((size_t)trials * 512);
    std::vector<float> C((size_t)trials * 256);
    srand(kmax);
    auto rnd = [](float s) { return s * ((float)rand() / RAND_MAX * 2.0f - 1.0f); };
    for (int t = 0; t < trials; ++t) {
      const float sa = (t & 1) ? 30000.0f : 2.0f, sc = (t & 2) ? 1000.0f : 0.5f;
      for (int r = 0; r < 16; ++r)
        for (int k = 0; k < 32; ++k) {
          A[(size_t)t * 512 + r * 32 + k] = k < kmax ? (_Float16)rnd(sa) : (_Float16)0.0f;
          B[(size_t)t * 512 + r * 32 + k] = k < kmax ? (_Float16)rnd(1.5f) : (_Float16)0.0f;
        }
      for (int e = 0; e < 256; ++e) C[(size_t)t * 256 + e] = (t & 4) ? rnd(sc) : 0.0f;
    }
    _Float16 *dA, *dB;
    float *dC, *d32, *d16;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4);
    hipMalloc(&d32, C.size() * 4); hipMalloc(&d16, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    compare_kernel<<<1, 64>>>(dA, dB, dC, d32, d16, trials);
    std::vector<float> r32(C.size()), r16(C.size());
    hipMemcpy(r32.data(), d32, C.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r16.data(), d16, C.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0;
    double worst = 0;
    for (size_t e = 0; e < C.size(); ++e)
      if (memcmp(&r32[e], &r16[e], 4) != 0) { ++diff; double d = fabs((double)r32[e] - r16[e]) / (fabs((double)r32[e]) + 1e-30); if (d > worst) worst = d; }
    printf("non-zero k < %2d: %zu of %zu outputs differ between 16x16x32 (high half zero) and 16x16x16 (worst relative %.3g)\n", kmax, diff,
           C.size(), worst);
  }
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  for (int wide = 0; wide < 2; ++wide) {
    for (int rep = 0; rep < 2; ++rep) {
      if (wide) timing_kernel<1><<<256, 256>>>(out, 20000); else timing_kernel<0><<<256, 256>>>(out, 20000);
      hipDeviceSynchronize();
    }
    float cyc;
    hipMemcpy(&cyc, out, 4, hipMemcpyDeviceToHost);
    printf("%s: %.2f cycles per MFMA per wave (one wave per SIMD, 8 independent accumulators)\n", wide ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_16x16x16_f16", cyc);
  }
  return 0;
}
