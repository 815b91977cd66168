// Stand-alone micro-benchmark of the discriminator GEMM variants (tile, BK, LDS stages, occupancy hint).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I humanoid_amp_amd/csrc tools/gemm_bench.hip \
//         humanoid_amp_amd/csrc/core.hip -o /tmp/gemm_bench && /tmp/gemm_bench [M] [N] [K]
// Prints us / TFLOP/s / fraction of the 157.3 TFLOP/s fp32-MFMA peak per variant, plus a checksum so variants can be
// compared for equality of results.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "disc_gemm.hpp"

using namespace amp;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("%s: %s\n", #x, hipGetErrorString(e));                            \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// Calibration: nothing but MFMAs on register operands (4 independent accumulators per wave), WPB waves per block.
template <int ACCS>
__global__ __launch_bounds__(256) void mfma_only_kernel(float* out, int iters) {
  floatx16 acc[ACCS];
#pragma unroll
  for (int i = 0; i < ACCS; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a += 1e-6f;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ACCS; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ACCS>
static void calib(float* out, int blocks_per_cu) {
  const int iters = 2048, grid = 256 * blocks_per_cu;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  mfma_only_kernel<ACCS><<<grid, 256>>>(out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  mfma_only_kernel<ACCS><<<grid, 256>>>(out, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double flops = (double)grid * 4 * iters * 4 * ACCS * 4096.0;
  printf("mfma-only: %d accumulators, %d blocks/CU: %.1f us  %.1f TF  %.3f of peak\n", ACCS, blocks_per_cu, ms * 1e3,
         flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
}

static double g_last_us = 0;

template <int BM_, int BN_, int BK_, int ST_, int MODE, int MW_>
static void run(const char* tag, GemmArgs g, int64_t M, int N, int K, float* out_dev, size_t out_floats) {
  g.n_tiles = N / BN_;
  g.m_tiles = (int)((M + BM_ - 1) / BM_);
  const unsigned grid = (unsigned)(((int64_t)g.m_tiles * g.n_tiles + 7) / 8 * 8);
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) disc_gemm_kernel<BM_, BN_, BK_, ST_, MODE, MW_><<<grid, kBlock>>>(g);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) disc_gemm_kernel<BM_, BN_, BK_, ST_, MODE, MW_><<<grid, kBlock>>>(g);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / reps;
  const double tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  std::vector<float> h(out_floats < 4096 ? out_floats : 4096);
  CK(hipMemcpy(h.data(), out_dev, h.size() * sizeof(float), hipMemcpyDeviceToHost));
  double cs = 0;
  for (float v : h) cs += v;
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, disc_gemm_kernel<BM_, BN_, BK_, ST_, MODE, MW_>, kBlock, 0));
  if (!getenv("QUIET"))
    printf("%-28s tile %3dx%3dx%2d stages %d mode %d  blocks/CU %d  %8.1f us  %6.1f TF  %.3f of peak   cs %.6e\n", tag, BM_, BN_,
           BK_, ST_, MODE, occ, us, tf, tf / 157.3, cs);
  fflush(stdout);
  g_last_us = us;
}

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536;
  const int N = argc > 2 ? atoi(argv[2]) : 512;
  const int K = argc > 3 ? atoi(argv[3]) : 1024;
  printf("GEMM M=%lld N=%d K=%d\n", (long long)M, N, K);
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hw3(N);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto& v : hA) v = rnd();
  for (auto& v : hW) v = rnd() * 0.1f;
  for (auto& v : hb) v = rnd();
  for (auto& v : hw3) v = rnd();
  float *A, *W, *b, *w3, *C, *P;
  CK(hipMalloc(&A, hA.size() * 4));
  CK(hipMalloc(&W, hW.size() * 4));
  CK(hipMalloc(&b, N * 4));
  CK(hipMalloc(&w3, N * 4));
  CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMalloc(&P, (size_t)M * 16 * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w3, hw3.data(), N * 4, hipMemcpyHostToDevice));
  GemmArgs g{};
  g.A = A; g.lda = K; g.M = M; g.K = K; g.W = W; g.Kp = K; g.bias = b; g.N = N; g.C = C; g.ldc = N; g.w3 = w3; g.partial = P;
  const size_t pf = (size_t)M * 4;
  if (getenv("CALIB")) {
    calib<4>(C, 1);
    calib<4>(C, 2);
    calib<1>(C, 4);
  }
#define V(BM, BN, BK, ST, MW) run<BM, BN, BK, ST, 1, MW>(#BM "x" #BN "x" #BK " s" #ST " w" #MW, g, M, N, K, P, pf)
#define V0(BM, BN, BK, ST, MW) run<BM, BN, BK, ST, 0, MW>(#BM "x" #BN "x" #BK " s" #ST " w" #MW " (store)", g, M, N, K, C, (size_t)M* N)
  // interleaved rounds: every variant sees the same thermal / clock state on average
  const int rounds = 6;
  std::vector<std::vector<double>> t(8);
  const char* names[8] = {"128x128x16 s1 w4", "128x128x32 s1 w3", "64x128x16 s1 w4", "64x128x32 s2 w2", "64x64x16 s1 w8",
                          "64x64x32 s1 w4", "64x64x32 s2 w4", "128x128x32 s2 w2"};
  setenv("QUIET", "1", 1);
  for (int r = 0; r < rounds; ++r) {
    int i = 0;
    if (N == 512) {
      V(128, 128, 16, 1, 4); t[i++].push_back(g_last_us);
      V(128, 128, 32, 1, 3); t[i++].push_back(g_last_us);
      V(64, 128, 16, 1, 4); t[i++].push_back(g_last_us);
      V(64, 128, 32, 2, 2); t[i++].push_back(g_last_us);
      V(64, 64, 16, 1, 8); t[i++].push_back(g_last_us);
      V(64, 64, 32, 1, 4); t[i++].push_back(g_last_us);
      V(64, 64, 32, 2, 4); t[i++].push_back(g_last_us);
      V(128, 128, 32, 2, 2); t[i++].push_back(g_last_us);
    } else {
      V0(128, 128, 16, 1, 4); t[i++].push_back(g_last_us);
      V0(128, 128, 32, 1, 3); t[i++].push_back(g_last_us);
      V0(64, 128, 16, 1, 4); t[i++].push_back(g_last_us);
      V0(64, 128, 32, 2, 2); t[i++].push_back(g_last_us);
      V0(64, 64, 16, 1, 8); t[i++].push_back(g_last_us);
      V0(64, 64, 32, 1, 4); t[i++].push_back(g_last_us);
      V0(64, 64, 32, 2, 4); t[i++].push_back(g_last_us);
      V0(128, 128, 32, 2, 2); t[i++].push_back(g_last_us);
    }
  }
  for (int i = 0; i < 8; ++i) {
    std::vector<double> v = t[i];
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2], tf = 2.0 * M * N * K / (med * 1e-6) / 1e12;
    printf("%-20s median %8.1f us (min %8.1f max %8.1f)  %6.1f TF  %.3f of peak\n", names[i], med, v.front(), v.back(), tf, tf / 157.3);
  }
  return 0;
}
