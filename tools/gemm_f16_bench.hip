// Stand-alone micro-benchmark + accuracy check of the split-operand fp16 GEMM (csrc/disc_gemm_f16.hpp).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I humanoid_amp_amd/csrc tools/gemm_f16_bench.hip \
//         humanoid_amp_amd/csrc/core.hip -o tools/bin/gemm_f16_bench && tools/bin/gemm_f16_bench [M] [N] [K]
// N == 512: mode 1 (partial logits); otherwise mode 0 (stores the planes of the hidden layer).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include <chrono>

#include "../tools/experiments/disc_gemm_f16_dma4.hpp"
#include "../tools/experiments/disc_gemm_f16_w4.hpp"
#include "../tools/experiments/disc_gemm_f16_panel.hpp"
#include "../tools/experiments/disc_gemm_f16_dma_xp.hpp"

using namespace amp;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("%s: %s\n", #x, hipGetErrorString(e));                            \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// Calibration: nothing but fp16 MFMAs on register operands (ACCS independent accumulators per wave).
template <int ACCS>
__global__ __launch_bounds__(256) void mfma_f16_only_kernel(float* out, int iters) {
  fx16 acc[ACCS];
#pragma unroll
  for (int i = 0; i < ACCS; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  h8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(blockIdx.x * 1e-4f + 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    a[0] += (_Float16)1e-3f;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ACCS; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Calibration 2: the same bare MFMA loop on operands that CHANGE from one MFMA to the next.  ENTROPY 0: eight register
// sets holding the same small constants (what mfma_f16_only_kernel measures); 1: eight sets of full-entropy random fp16
// values (what a GEMM on real data feeds the multipliers).  The matrix pipe's power -- and with it the clock the chip
// sustains -- depends on how many multiplier inputs toggle between consecutive MFMAs.
template <int ENTROPY>
__global__ __launch_bounds__(256, 1) void mfma_f16_entropy_kernel(float* out, int iters) {
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
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48; ++u)
      acc[u & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 7], b[(u * 3 + (u >> 3)) & 7], acc[u & 15], 0, 0, 0);
  }
  float t = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int ENTROPY>
static void calib_entropy(float* out, int iters) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  mfma_f16_entropy_kernel<ENTROPY><<<256, 256>>>(out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  mfma_f16_entropy_kernel<ENTROPY><<<256, 256>>>(out, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double flops = 256.0 * 4 * iters * 48 * 32768.0;
  printf("fp16 MFMA only, 16 accumulators, one wave per SIMD, %s operands, %d x 48 MFMAs: %.1f us  %.1f TF  %.3f of the 2516.8 TF peak\n",
         ENTROPY ? "RANDOM (8 rotating register sets)" : "constant", iters, ms * 1e3, flops / (ms * 1e-3) / 1e12,
         flops / (ms * 1e-3) / 1e12 / 2516.8);
}

// Store-only floor of layer 1: every thread writes 16 B per plane the way the epilogue does (row segments of 128 B).
__global__ __launch_bounds__(256) void store_only_kernel(_Float16* out, int64_t rows, int cols, int64_t plane) {
  const int64_t chunks_per_row = cols / 8, total = rows * chunks_per_row;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (int64_t)gridDim.x * 256) {
    h8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)(float)(c & 1023);
    *reinterpret_cast<h8*>(out + c * 8) = v;
    *reinterpret_cast<h8*>(out + plane + c * 8) = v;
  }
}

// The same bytes in the GEMM epilogue's ORDER: a workgroup writes a 256-row x 1-KB column slab of the block-layout
// hidden layer (row pitch `pitch` bytes), a wave one 1-KB row segment per instruction.
__global__ __launch_bounds__(256) void store_tiles_kernel(unsigned char* out, int64_t rows, int64_t pitch) {
  const int slabs = (int)(pitch / 1024);
  const int64_t mt = blockIdx.x / slabs;
  const int nt = blockIdx.x % slabs;
  if (mt * 256 >= rows) return;
  uv4 v;
  v[0] = v[1] = v[2] = v[3] = threadIdx.x;
  unsigned char* base = out + mt * 256 * pitch + (int64_t)nt * 1024 + (threadIdx.x & 63) * 16;
#pragma unroll 8
  for (int i = 0; i < 64; ++i) *reinterpret_cast<uv4*>(base + (int64_t)(i * 4 + (threadIdx.x >> 6)) * pitch) = v;
}

// Fill-path probe: one 512-thread workgroup per CU streams its 256-row A slab and the shared 256-row B slab into
// LDS by LDS-DMA exactly like disc_gemm_f16_dma_kernel (4 x 32-KB ring, counted vmcnt, one barrier per k-step) but
// computes nothing.  SEG = bytes a piece takes from one row: 32 (k-step 16), 64 or 128.
template <int SEG>
__global__ __launch_bounds__(512, 1) void fill_probe_kernel(const _Float16* A, int64_t lda, int64_t plane_a, const _Float16* W, int Kp,
                                                            int64_t plane_w, int m_tiles, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per_xcd = (m_tiles * 2 + 7) / 8;
  const int v = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (v >= m_tiles * 2) return;
  const int mt = v >> 1, nt = v & 1;
  constexpr int ROWS = 1024 / SEG;            // rows per piece
  constexpr int CH = SEG / 16;                // 16-B chunks per row segment
  constexpr int KSTEP = SEG / 2;              // halves per row segment = k extent of one ring slot
  // ring slot = 4 planes x 256 rows x SEG bytes; pieces per plane = 256 / ROWS; 8 waves share them
  constexpr int PPP = 256 / ROWS, PPW = 4 * PPP / 8;  // pieces per plane / per wave per slot
  const int nslots = Kp / KSTEP;
  auto issue = [&](int p, int slot) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int q = j * 8 + wave, plane = q / PPP, piece = q % PPP;
      const int r = piece * ROWS + lane / CH, c = lane % CH;
      const _Float16* src = plane < 2 ? A + plane * plane_a + ((int64_t)mt * 256 + r) * lda + p * KSTEP + 8 * c
                                      : W + (plane - 2) * plane_w + ((int64_t)nt * 256 + r) * Kp + p * KSTEP + 8 * c;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + (slot % (131072 / (1024 * SEG))) * (1024 * SEG) + q * 1024), 16, 0, 0);
    }
  };
  constexpr int RING = 131072 / (1024 * SEG);  // slots in 128 KB
  for (int p = 0; p < RING - 1 && p < nslots; ++p) issue(p, p);
  for (int p = 0; p < nslots; ++p) {
    if (p + RING - 1 < nslots) issue(p + RING - 1, p + RING - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (tid == 0) sink[blockIdx.x] = reinterpret_cast<float*>(lds)[lane];
}

template <int SEG>
static void fill_probe(const _Float16* Ap, int64_t M, int K, const _Float16* Wp, int N, float* sink) {
  const int m_tiles = (int)(M / 256);
  const unsigned grid = (unsigned)((m_tiles * 2 + 7) / 8 * 8);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_probe_kernel<SEG>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  fill_probe_kernel<SEG><<<grid, 512, 131072>>>(Ap, K, M * K, Wp, K, (int64_t)N * K, m_tiles, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) fill_probe_kernel<SEG><<<grid, 512, 131072>>>(Ap, K, M * K, Wp, K, (int64_t)N * K, m_tiles, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)m_tiles * 2 * 2.0 * 256 * K * 2 * 2;  // per workgroup: A slab + B slab, 2 planes each
  printf("fill probe, %3d B per row segment: %8.1f us  %.2f TB/s into LDS (%.0f MB)\n", SEG, ms * 200, bytes / (ms / 5 * 1e-3) / 1e12, bytes / 1e6);
}

template <int ACCS>
static void calib(float* out, int blocks_per_cu, int iters) {
  const int grid = 256 * blocks_per_cu;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  mfma_f16_only_kernel<ACCS><<<grid, 256>>>(out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  mfma_f16_only_kernel<ACCS><<<grid, 256>>>(out, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double flops = (double)grid * 4 * iters * 4 * ACCS * 32768.0;
  printf("fp16 mfma-only: %d accumulators, %d blocks/CU, %d iters: %.1f us  %.1f TF  %.3f of the 2516.8 TF peak\n", ACCS, blocks_per_cu,
         iters, ms * 1e3, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 2516.8);
}

static double g_us = 0;

template <int TM, int TN, int BK, int MODE, int MW>
static void run(GemmF16Args g, int64_t M, int N, int K, bool quiet) {
  constexpr int PF = 1;
  g.n_tiles = N / (64 * TN);
  g.m_tiles = (int)((M + 64 * TM - 1) / (64 * TM));
  const unsigned grid = (unsigned)(((int64_t)g.m_tiles * g.n_tiles + 7) / 8 * 8);
  constexpr int lds = gemm_f16_lds_bytes<TM, TN, BK>();
  auto kern = disc_gemm_f16_kernel<TM, TN, BK, MODE, MW>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) kern<<<grid, kBlock, lds>>>(g);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) kern<<<grid, kBlock, lds>>>(g);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  g_us = ms * 1e3 / reps;
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kBlock, lds));
  if (!quiet) {
    const double tf = 2.0 * M * N * K / (g_us * 1e-6) / 1e12;
    printf("tile %3dx%3dx%2d mode %d minw %d pf %d lds %6d blocks/CU %d  %8.1f us  %6.1f TF(alg)  %.3f of fp16 peak executed\n", 64 * TM,
           64 * TN, BK, MODE, MW, PF, lds, occ, g_us, tf, 3 * tf / 2516.6);
    fflush(stdout);
  }
}

static _Float16 *g_Ab = nullptr, *g_Wb = nullptr;
static bool g_dma_last = false;  // the last kernel run wrote H in block layout (check() must index accordingly)

template <int MODE, int XP = 0, int TM = 4, int TN = 2>
static void run_dma(GemmF16Args g, int64_t M, int N, int K, bool quiet) {
  using T = DmaTile<TM, TN>;
  if (MODE == 1) g.A = g_Ab;  // block layout (MODE 0: g.A already is)
  g.W = g_Wb;
  g_dma_last = true;
  g.n_tiles = N / T::BN;
  g.m_tiles = (int)((M + T::BM - 1) / T::BM);
  unsigned grid = (unsigned)(((int64_t)g.m_tiles * g.n_tiles + 7) / 8 * 8);
  // layer 1 of the product: persistent workgroups, one per LDS slot (NOPERSIST=1: one workgroup per tile, as before round 3)
  if (MODE == 0 && XP == 0 && !getenv("NOPERSIST")) grid = std::min(grid, 256u * T::kWgPerCu);
  // XP = 0: the PRODUCT kernel; XP != 0: the ablation copy under tools/experiments (round-1 epilogue)
  void (*kern)(GemmF16Args) = disc_gemm_f16_dma_xp_kernel<MODE, XP, TM, TN>;
  if (XP == 0) kern = disc_gemm_f16_dma_kernel<MODE, TM, TN>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::kLds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) kern<<<grid, kDmaThreads, T::kLds>>>(g);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) kern<<<grid, kDmaThreads, T::kLds>>>(g);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  g_us = ms * 1e3 / reps;
  if (!quiet) {
    const double tf = 2.0 * M * N * K / (g_us * 1e-6) / 1e12;
    printf("LDS-DMA %3dx%3dx32 blocks, 2 stages, 512 thr            %8.1f us   %6.1f TF(alg)  %.3f of fp16 peak executed\n", T::BM, T::BN, g_us, tf,
           3 * tf / 2516.6);
    fflush(stdout);
  }
}

template <int MODE, int NA, int NW>
static void run_dma4(GemmF16Args g, int64_t M, int N, int K, bool quiet) {
  if (MODE == 1) g.A = g_Ab;  // block layout (MODE 0: g.A already is)
  g.W = g_Wb;
  g_dma_last = true;
  g.n_tiles = N / kDma4BN;
  g.m_tiles = (int)((M + kDma4BM - 1) / kDma4BM);
  const unsigned grid = (unsigned)(((int64_t)g.m_tiles * g.n_tiles + 7) / 8 * 8);
  constexpr int lds = dma4_lds_bytes<NA, NW>();
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gemm_f16_dma4_kernel<MODE, NA, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) disc_gemm_f16_dma4_kernel<MODE, NA, NW><<<grid, kDma4Threads, lds>>>(g);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) disc_gemm_f16_dma4_kernel<MODE, NA, NW><<<grid, kDma4Threads, lds>>>(g);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  g_us = ms * 1e3 / reps;
  if (!quiet) {
    const double tf = 2.0 * M * N * K / (g_us * 1e-6) / 1e12;
    printf("LDS-DMA4 128x128x32 blocks, %d + %d stages, 256 thr (4 waves of 64x64) %8.1f us   %6.1f TF(alg)  %.3f of fp16 peak executed\n", NA, NW, g_us, tf,
           3 * tf / 2516.6);
    fflush(stdout);
  }
}

static void run_w4(GemmF16Args g, int64_t M, int N, int K, bool quiet) {
  g.A = g_Ab;  // block layout
  g.W = g_Wb;
  g_dma_last = true;
  g.n_tiles = N / kW4BN;
  g.m_tiles = (int)((M + kW4BM - 1) / kW4BM);
  const unsigned grid = (unsigned)(((int64_t)g.m_tiles * g.n_tiles + 7) / 8 * 8);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gemm_f16_w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kW4LdsBytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) disc_gemm_f16_w4_kernel<<<grid, kW4Threads, kW4LdsBytes>>>(g);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) disc_gemm_f16_w4_kernel<<<grid, kW4Threads, kW4LdsBytes>>>(g);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  g_us = ms * 1e3 / reps;
  if (!quiet) {
    const double tf = 2.0 * M * N * K / (g_us * 1e-6) / 1e12;
    printf("W4 256x256x32 blocks, 2 stages, 256 thr (4 waves of 128x128)  %8.1f us   %6.1f TF(alg)  %.3f of fp16 peak executed\n", g_us, tf,
           3 * tf / 2516.6);
    fflush(stdout);
  }
}

template <int KS, int TM>
static void run_panel(GemmF16Args g, int64_t M, int N, int K, bool quiet) {
  g.W = g_Wb;  // block layout weights; the activations (g.A) are in block layout already
  g_dma_last = true;
  // operand extents the kernel's indexing assumes, checked on the host before anything is launched: whole 128-row panels
  // (the hidden layer here has NO row padding: M * N * 4 bytes), whole 128-column chunks, the reduction inside KS k-steps
  if (M % kPanelBM != 0 || N % kPanelCH != 0 || N > kPanelMaxN || K > 16 * KS || g.ldh != N || g.lda != K || g.Kp != K) {
    printf("run_panel: shapes outside what the panel kernel indexes (M %% 128, N %% 128, N <= 2048, K <= %d)\n", 16 * KS);
    exit(1);
  }
  const unsigned grid = (unsigned)((M + kPanelBM - 1) / kPanelBM);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gemm_f16_panel_kernel<KS, TM>), hipFuncAttributeMaxDynamicSharedMemorySize, kPanelLdsBytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) disc_gemm_f16_panel_kernel<KS, TM><<<grid, panel_threads<TM>(), kPanelLdsBytes>>>(g);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) disc_gemm_f16_panel_kernel<KS, TM><<<grid, panel_threads<TM>(), kPanelLdsBytes>>>(g);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  g_us = ms * 1e3 / reps;
  if (!quiet) {
    const double tf = 2.0 * M * N * K / (g_us * 1e-6) / 1e12;
    printf("PANEL 128 rows x N, %2d k-steps, 8-stage weight ring, %d waves of %dx64 per chunk  %8.1f us   %6.1f TF(alg)  %.3f of fp16 peak executed\n",
           KS, panel_threads<TM>() / 64, 32 * TM, g_us, tf, 3 * tf / 2516.6);
    fflush(stdout);
  }
}

// CONC=1 (with W4=1, mode 1): does a streaming kernel (the env step's byte mix: 78 MB read + 158 MB written, ~16 VGPRs, no
// LDS -- it fits beside either GEMM kernel on a CU) run UNDER a layer-2 GEMM on a second stream, or do the two add up?
typedef float conc_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void conc_mix_kernel(const conc_f4* __restrict__ src, conc_f4* __restrict__ dst, long n_items) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_items; i += stride) {
    const conc_f4 v = src[i];
    dst[i] = v;
    dst[n_items + i] = v * 2.0f;
  }
}
static void run_conc(GemmF16Args g, int64_t M, int N, int K) {
  GemmF16Args g8 = g, g4 = g;
  g8.A = g_Ab; g8.W = g_Wb; g8.n_tiles = N / 256; g8.m_tiles = (int)((M + 255) / 256);
  g4 = g8; g4.n_tiles = N / kW4BN; g4.m_tiles = (int)((M + kW4BM - 1) / kW4BM);
  const unsigned grid8 = (unsigned)(((int64_t)g8.m_tiles * g8.n_tiles + 7) / 8 * 8);
  const unsigned grid4 = (unsigned)(((int64_t)g4.m_tiles * g4.n_tiles + 7) / 8 * 8);
  void (*k8)(GemmF16Args) = disc_gemm_f16_dma_kernel<1, 4, 2>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, DmaTile<4, 2>::kLds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gemm_f16_w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kW4LdsBytes));
  const long n_items = (long)(78e6 / 16);
  conc_f4 *src[3], *dst[3];
  for (int i = 0; i < 3; ++i) {
    CK(hipMalloc(&src[i], n_items * 16)); CK(hipMalloc(&dst[i], 2 * n_items * 16));
    CK(hipMemset(src[i], 0, n_items * 16)); CK(hipMemset(dst[i], 0, 2 * n_items * 16));
  }
  hipStream_t s2;
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int reps = 12;
  auto timed = [&](const char* label, int gemm, bool mix) {
    for (int warm = 0; warm < 2; ++warm) {
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < reps; ++i) {
        if (gemm == 8) k8<<<grid8, kDmaThreads, DmaTile<4, 2>::kLds>>>(g8);
        if (gemm == 4) disc_gemm_f16_w4_kernel<<<grid4, kW4Threads, kW4LdsBytes>>>(g4);
        if (mix) conc_mix_kernel<<<2048, 256, 0, s2>>>(src[i % 3], dst[i % 3], n_items);
      }
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      if (warm) printf("%-58s %8.1f us per iteration\n", label, us);
    }
  };
  timed("streaming kernel alone (236 MB)", 0, true);
  timed("8-wave layer-2 GEMM alone", 8, false);
  timed("8-wave layer-2 GEMM + streaming kernel on a second stream", 8, true);
  timed("4-wave layer-2 GEMM alone", 4, false);
  timed("4-wave layer-2 GEMM + streaming kernel on a second stream", 4, true);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536;
  const int N = argc > 2 ? atoi(argv[2]) : 512;
  const int K = argc > 3 ? atoi(argv[3]) : 1024;
  const int mode = N == 512 ? 1 : 0;
  if (K % 32 != 0 || N % 256 != 0 || M % 256 != 0) { puts("need K % 32 == 0, N % 256 == 0, M % 256 == 0 (the kernels index whole k-blocks / tiles)"); return 1; }
  printf("f16-split GEMM M=%lld N=%d K=%d mode %d\n", (long long)M, N, K, mode);
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hw3(N);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto& v : hA) v = std::max(rnd() * 3.0f, 0.0f) * (1.0f + rnd());
  for (int64_t m = 5; m < M; m += M / 64) {  // rows far below the tensor's bound: fp16 subnormal territory
    const float tiny = m % 2 ? 3e-7f : 1e-9f;
    for (int k = 0; k < K; ++k) hA[m * K + k] *= tiny;
  }
  for (auto& v : hW) v = rnd() * 0.1f;
  for (auto& v : hb) v = rnd();
  for (auto& v : hw3) v = rnd();
  float amax = 0, wmax = 0;
  for (float v : hA) amax = std::max(amax, std::fabs(v));
  for (float v : hW) wmax = std::max(wmax, std::fabs(v));
  const float sa = plane_scale(amax), sw = plane_scale(wmax);
  // range record such that layer_scales() reproduces these scales: bound_x = amax (clip), |H| <= 0 * bound_x + 100
  DiscRange hr{};
  hr.s_w1 = hr.s_w2 = sw; hr.wsum1 = 0.0f; hr.bmax1 = mode == 0 ? 100.0f : amax; hr.clip = amax;
  const float sh = plane_scale(100.0f);
  float *A, *W, *b, *w3, *P, *one;
  DiscRange* sc;
  _Float16 *Ap, *Wp, *Hp, *Ab, *Wb;  // planar (register-staged kernels) and block layout (LDS-DMA kernels)
  CK(hipMalloc(&A, hA.size() * 4));
  CK(hipMalloc(&W, hW.size() * 4));
  CK(hipMalloc(&Ap, hA.size() * 4));
  CK(hipMalloc(&Wp, hW.size() * 4));
  CK(hipMalloc(&Hp, (size_t)M * N * 4));
  CK(hipMalloc(&Ab, hA.size() * 4));
  CK(hipMalloc(&Wb, hW.size() * 4));
  CK(hipMalloc(&b, N * 4));
  CK(hipMalloc(&w3, N * 4));
  CK(hipMalloc(&P, (size_t)M * 16 * 4));
  CK(hipMalloc(&sc, sizeof(DiscRange)));
  CK(hipMalloc(&one, 8));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w3, hw3.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, &hr, sizeof(DiscRange), hipMemcpyHostToDevice));
  float s2[2] = {sa, sw};
  CK(hipMemcpy(one, s2, 8, hipMemcpyHostToDevice));
  if (mode == 0) split_rows_blocks_kernel<<<(unsigned)((M * K / 4 + 255) / 256), 256>>>(A, M, K, K, one, Ap, K);  // the scaled input: block layout
  else split_rows_f16_kernel<<<(unsigned)((M * K / 4 + 255) / 256), 256>>>(A, M, K, K, one, Ap, K, M * K);
  split_rows_f16_kernel<<<(unsigned)(((int64_t)N * K / 4 + 255) / 256), 256>>>(W, N, K, K, one + 1, Wp, K, (int64_t)N * K);
  if (mode == 1) split_rows_blocks_kernel<<<(unsigned)((M * K / 4 + 255) / 256), 256>>>(A, M, K, K, one, Ab, K);
  split_rows_blocks_kernel<<<(unsigned)(((int64_t)N * K / 4 + 255) / 256), 256>>>(W, N, K, K, one + 1, Wb, K);
  CK(hipDeviceSynchronize());
  GemmF16Args g{};
  g.A = Ap; g.lda = K; g.plane_a = M * K; g.M = M; g.W = Wp; g.plane_w = (int64_t)N * K; g.Kp = K; g.N = N;
  g.bias = b; g.range = sc; g.amax = nullptr; g.layer = mode == 0 ? 1 : 2; g.H = Hp; g.ldh = N; g.plane_h = M * N; g.w3 = w3; g.partial = P;

  g_Ab = Ab; g_Wb = Wb;
  auto check = [&](int n_tiles) {
    if (mode == 1) n_tiles = N / 32;  // canonical partial logits: one per (row, 32-column block), whatever the tile
    const bool blocks = g_dma_last;
    g_dma_last = false;
    const int rows = 64;
    double worst = 0, scale = 0;
    if (mode == 1) {
      std::vector<float> hp((size_t)M * n_tiles);
      CK(hipMemcpy(hp.data(), P, hp.size() * 4, hipMemcpyDeviceToHost));
      for (int r = 0; r < rows; ++r) {
        const int64_t m = (int64_t)r * (M / rows) + (r % 7);
        double ref = 0, got = 0;
        for (int n = 0; n < N; ++n) {
          double d = hb[n];
          for (int k = 0; k < K; ++k) d += (double)hA[m * K + k] * hW[(size_t)n * K + k];
          ref += std::max(d, 0.0) * hw3[n];
        }
        for (int t = 0; t < n_tiles; ++t) got += hp[m * n_tiles + t];
        worst = std::max(worst, std::fabs(got - ref));
        scale = std::max(scale, std::fabs(ref));
      }
    } else {
      std::vector<_Float16> hh((size_t)2 * M * N);
      CK(hipMemcpy(hh.data(), Hp, hh.size() * 2, hipMemcpyDeviceToHost));
      for (int r = 0; r < rows; ++r) {
        const int64_t m = (int64_t)r * (M / rows) + (r % 7);
        for (int n = 0; n < N; n += 3) {
          double d = hb[n];
          for (int k = 0; k < K; ++k) d += (double)hA[m * K + k] * hW[(size_t)n * K + k];
          const double ref = std::max(d, 0.0);
          const size_t o0 = blocks ? (size_t)m * 2 * N + (size_t)(n >> 5) * 64 + (n & 31) : (size_t)m * N + n;
          const size_t o1 = blocks ? o0 + 32 : (size_t)M * N + m * N + n;
          const double got = ((double)(float)hh[o0] + (double)(float)hh[o1]) / sh;
          worst = std::max(worst, std::fabs(got - ref));
          scale = std::max(scale, std::fabs(ref));
        }
      }
    }
    printf("    max |err| vs fp64 on %d rows: %.3e (|ref| up to %.3e)\n", rows, worst, scale);
    if (mode == 1) {
      std::vector<float> hp((size_t)M * n_tiles);
      CK(hipMemcpy(hp.data(), P, hp.size() * 4, hipMemcpyDeviceToHost));
      for (int64_t m = 5; m < M; m += M / 64 * 21) {
        double ref = 0, got = 0, refb = 0;
        for (int n = 0; n < N; ++n) {
          double d = hb[n];
          for (int k = 0; k < K; ++k) d += (double)hA[m * K + k] * hW[(size_t)n * K + k];
          ref += std::max(d, 0.0) * hw3[n];
          refb += std::max((double)hb[n], 0.0) * hw3[n];
        }
        for (int t = 0; t < n_tiles; ++t) got += hp[m * n_tiles + t];
        printf("    tiny row %lld: ref %.9e got %.9e (bias-only %.9e)\n", (long long)m, ref, got, refb);
      }
    }
  };

#define V(TM, TN, BK, MW, PF)                                        \
  do {                                                               \
    if (PF == 1 && mode == 1) run<TM, TN, BK, 1, MW>(g, M, N, K, quiet);    \
    else if (PF == 1) run<TM, TN, BK, 0, MW>(g, M, N, K, quiet);     \
  } while (0)
  bool quiet = false;
  if (getenv("ENTROPY")) {  // what the matrix pipe sustains on constant vs random operands, 50 us .. 1 ms bursts
    for (int iters : {64, 256, 1024}) {
      calib_entropy<0>((float*)Hp, iters);
      calib_entropy<1>((float*)Hp, iters);
    }
    calib_entropy<0>((float*)Hp, 256);
    calib_entropy<1>((float*)Hp, 256);
    return 0;
  }
  if (getenv("CALIB")) {
    if (N != 1024) { puts("CALIB needs N = 1024 (the store probes write M x 1024 x 2 planes)"); return 1; }
    calib<4>((float*)Hp, 1, 512);    // ~50 us
    calib<4>((float*)Hp, 1, 2048);   // ~200 us
    calib<4>((float*)Hp, 2, 2048);
    calib<4>((float*)Hp, 1, 16384);  // ~1.7 ms
    for (int grid : {256, 512, 1024, 2048, 8192, 65536}) {  // 256 threads each: 256 -> one 4-wave workgroup per CU
      hipEvent_t a, b;
      CK(hipEventCreate(&a));
      CK(hipEventCreate(&b));
      store_only_kernel<<<grid, 256>>>(Hp, M, 1024, M * 1024);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      for (int i = 0; i < 5; ++i) store_only_kernel<<<grid, 256>>>(Hp, M, 1024, M * 1024);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      printf("store-only %lld x 1024 x 2 planes (%.0f MB), grid %d: %.1f us  %.2f TB/s\n", (long long)M, M * 4096.0 / 1e6, grid, ms * 200,
             M * 4096.0 / (ms / 5 * 1e-3) / 1e12);
    }
    {
      hipEvent_t a, b;
      CK(hipEventCreate(&a));
      CK(hipEventCreate(&b));
      const unsigned grid = (unsigned)((M + 255) / 256 * 4);
      store_tiles_kernel<<<grid, 256>>>((unsigned char*)Hp, M, 4096);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      for (int i = 0; i < 5; ++i) store_tiles_kernel<<<grid, 256>>>((unsigned char*)Hp, M, 4096);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      printf("store-only, tile order (256-row x 1-KB slabs, 4-KB row pitch), grid %u: %.1f us  %.2f TB/s\n", grid, ms * 200,
             M * 4096.0 / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
  }
  if (getenv("FILL") && mode == 1) {
    fill_probe<32>(Ap, M, K, Wp, N, P);
    fill_probe<64>(Ap, M, K, Wp, N, P);
    fill_probe<128>(Ap, M, K, Wp, N, P);
    fill_probe<32>(Ap, M, K, Wp, N, P);
    return 0;
  }
  if (getenv("SMALL")) {  // rectangular register-staged tiles for shards that 64 x 64 tiles spread thin
    for (int r = 0; r < 2; ++r) {
      V(1, 1, 64, 4, 1); check(N / 64);
      V(2, 1, 64, 2, 1); check(N / 64);
      V(1, 2, 64, 2, 1); check(N / 128);
      V(2, 1, 32, 3, 1); check(N / 64);
      V(1, 2, 32, 3, 1); check(N / 128);
      V(2, 1, 64, 3, 1); check(N / 64);
    }
    return 0;
  }
  if (getenv("PANEL") && mode == 0) {
    for (int rep = 0; rep < 3; ++rep) {
      if (K == 192) { run_panel<12, 1>(g, M, N, K, false); check(0); }
      run_dma<0, 0, 4, 2>(g, M, N, K, false); check(0);
    }
    return 0;
  }
  if (getenv("W4") && getenv("CONC") && mode == 1) {
    run_conc(g, M, N, K);
    return 0;
  }
  if (getenv("W4")) {
    for (int rep = 0; rep < 3; ++rep) {
      if (mode == 0) {
        run_dma<0, 0, 4, 2>(g, M, N, K, false); check(0);
      } else {
        run_w4(g, M, N, K, false); check(0);
        run_dma<1, 0, 4, 2>(g, M, N, K, false); check(0);
      }
    }
    return 0;
  }
  if (getenv("DMA4")) {
    for (int rep = 0; rep < 3; ++rep) {
      if (mode == 0) {
        run_dma4<0, 3, 3>(g, M, N, K, false); check(0);
        run_dma4<0, 5, 3>(g, M, N, K, false); check(0);
        run_dma4<0, 6, 3>(g, M, N, K, false); check(0);
        run_dma4<0, 4, 4>(g, M, N, K, false); check(0);
        run_dma<0, 0, 2, 1>(g, M, N, K, false); check(0);
      } else {
        run_dma4<1, 3, 3>(g, M, N, K, false); check(0);
        run_dma4<1, 5, 3>(g, M, N, K, false); check(0);
        run_dma4<1, 6, 3>(g, M, N, K, false); check(0);
        run_dma4<1, 4, 4>(g, M, N, K, false); check(0);
        run_dma<1, 0, 2, 1>(g, M, N, K, false); check(0);
      }
    }
    return 0;
  }
  if (getenv("FREE")) {  // free-running waves (XP = 5: one barrier per k-block, no ping-pong) on the small tiles
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) {
        run_dma<0, 5, 2, 1>(g, M, N, K, false); check(N / 128);
        run_dma<0, 5, 1, 1>(g, M, N, K, false); check(N / 128);
        run_dma<0, 5, 2, 2>(g, M, N, K, false); check(N / 256);
      } else {
        run_dma<1, 5, 2, 1>(g, M, N, K, false); check(N / 128);
        run_dma<1, 5, 1, 1>(g, M, N, K, false); check(N / 128);
        run_dma<1, 5, 2, 2>(g, M, N, K, false); check(N / 256);
        run_dma<1, 5, 4, 1>(g, M, N, K, false); check(N / 128);
      }
    }
    return 0;
  }
  if (getenv("CONC0") && mode == 0) {
    // Layer 1 WITHOUT its global stores (ablation XP = 3 of the round-1-epilogue copy: everything else kept) next to a kernel that
    // does nothing but store the hidden layer's bytes, on a second stream: do a compute phase and a store phase overlap AT ALL
    // on this chip when they are not the same workgroups?
    using T = DmaTile<4, 2>;
    GemmF16Args gx = g;
    gx.W = g_Wb;
    gx.n_tiles = N / T::BN;
    gx.m_tiles = (int)((M + T::BM - 1) / T::BM);
    const unsigned gridx = (unsigned)(((int64_t)gx.m_tiles * gx.n_tiles + 7) / 8 * 8);
#ifdef AMP_L1_NO_STORES
    void (*kx)(GemmF16Args) = disc_gemm_f16_dma_kernel<0, 4, 2>;  // the PRODUCT kernel compiled without its stores, one workgroup per tile
    puts("(product kernel built with -DAMP_L1_NO_STORES)");
#else
    void (*kx)(GemmF16Args) = disc_gemm_f16_dma_xp_kernel<0, 3, 4, 2>;  // round-1-epilogue copy: no stores, no fills, no fragment reads
    puts("(ablation copy XP = 3: no stores AND no fills / fragment reads in the loop)");
#endif
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kx), hipFuncAttributeMaxDynamicSharedMemorySize, T::kLds));
    unsigned char* sink;
    CK(hipMalloc(&sink, (size_t)M * 4096));
    const unsigned grids = (unsigned)((M + 255) / 256 * 4);
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int reps = 12;
    auto timed = [&](const char* label, bool gemm, bool stores) {
      for (int warm = 0; warm < 2; ++warm) {
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) {
          if (gemm) kx<<<gridx, kDmaThreads, T::kLds>>>(gx);
          if (stores) store_tiles_kernel<<<grids, 256, 0, s2>>>(sink, M, 4096);
        }
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        if (warm) printf("%-70s %8.1f us per iteration\n", label, us);
      }
    };
    timed("layer 1 without its global stores, alone", true, false);
    timed("store-only kernel (the hidden layer's bytes), alone", false, true);
    timed("both, on two streams", true, true);
    run_dma<0, 0, 4, 2>(g, M, N, K, false);
    return 0;
  }
#ifdef AMP_DMA_TIMELINE
  if (getenv("TIMELINE") && mode == 0) {  // per-workgroup phase stamps of ONE layer-1 launch (100 MHz wall clock: 10 ns ticks)
    using T = DmaTile<4, 2>;
    g.W = g_Wb;
    g.n_tiles = N / T::BN;
    g.m_tiles = (int)((M + T::BM - 1) / T::BM);
    const unsigned tiles = (unsigned)(((int64_t)g.m_tiles * g.n_tiles + 7) / 8 * 8);
    const unsigned launch = getenv("NOPERSIST") ? tiles : std::min(tiles, 256u);
    const unsigned grid = tiles;  // stamp rows
    unsigned long long* tl;
    CK(hipMalloc(&tl, (size_t)grid * 64));
    CK(hipMemset(tl, 0, (size_t)grid * 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gemm_f16_dma_kernel<0, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, T::kLds));
    for (int i = 0; i < 3; ++i) disc_gemm_f16_dma_kernel<0, 4, 2><<<launch, kDmaThreads, T::kLds>>>(g);
    CK(hipDeviceSynchronize());
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_dma_timeline), &tl, sizeof(tl)));
    disc_gemm_f16_dma_kernel<0, 4, 2><<<launch, kDmaThreads, T::kLds>>>(g);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)grid * 8);
    CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (unsigned b = 0; b < grid; ++b) if (h[b * 8]) t0 = std::min(t0, h[b * 8]);
    double sum[5] = {0, 0, 0, 0, 0};
    int n = 0;
    for (unsigned b = 0; b < grid; ++b) {
      if (!h[b * 8 + 4]) continue;
      ++n;
      for (int i = 0; i < 4; ++i) sum[i] += (double)(h[b * 8 + i + 1] - h[b * 8 + i]) * 0.01;
      sum[4] += (double)(h[b * 8 + 4] - t0) * 0.01;
    }
    printf("timeline over %d workgroups (us): start->k-loop %.2f | k-loop %.2f | epilogue issue %.2f | store drain %.2f | mean end %.2f\n", n,
           sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n);
    for (unsigned b = 0; b < grid; b += grid / 16) {
      const unsigned hw = (unsigned)h[b * 8 + 5];
      printf("  wg %4u cu %2u se %u xcc?: start %7.2f  kloop %7.2f  kend %7.2f  issued %7.2f  done %7.2f\n", b, (hw >> 8) & 15, (hw >> 13) & 7,
             (h[b * 8] - t0) * 0.01, (h[b * 8 + 1] - t0) * 0.01, (h[b * 8 + 2] - t0) * 0.01, (h[b * 8 + 3] - t0) * 0.01, (h[b * 8 + 4] - t0) * 0.01);
    }
    // per-CU view: the workgroups that ran on the same (xcd = b & 7, hw id) in start order
    return 0;
  }
#endif
  if (getenv("TILES")) {
    for (int rep = 0; rep < 3; ++rep) {
      if (mode == 0) {
        run_dma<0, 0, 4, 2>(g, M, N, K, false); check(N / 256);
        run_dma<0, 0, 2, 2>(g, M, N, K, false); check(N / 256);
        run_dma<0, 0, 2, 1>(g, M, N, K, false); check(N / 128);
        run_dma<0, 0, 4, 1>(g, M, N, K, false); check(N / 128);
        run_dma<0, 0, 1, 1>(g, M, N, K, false); check(N / 128);
      } else {
        run_dma<1, 0, 4, 2>(g, M, N, K, false); check(N / 256);
        run_dma<1, 0, 2, 2>(g, M, N, K, false); check(N / 256);
        run_dma<1, 0, 2, 1>(g, M, N, K, false); check(N / 128);
        run_dma<1, 0, 4, 1>(g, M, N, K, false); check(N / 128);
        run_dma<1, 0, 1, 1>(g, M, N, K, false); check(N / 128);
      }
    }
    return 0;
  }
  if (getenv("XPS") && mode == 1) {  // the same ablations on the 128 x 128 tile of the 8 192-row shards (TM = 2, TN = 1)
    for (int rep = 0; rep < 2; ++rep) {
      run_dma<1, 0, 2, 1>(g, M, N, K, false); puts("  ^ 128x128 LDS-DMA kernel, full");
      run_dma<1, 1, 2, 1>(g, M, N, K, false); puts("  ^ no fills in the loop");
      run_dma<1, 2, 2, 1>(g, M, N, K, false); puts("  ^ no fills, no fragment reads: barriers + MFMA");
      run_dma<1, 5, 2, 1>(g, M, N, K, false); puts("  ^ free-running waves, one barrier per k-block");
    }
    return 0;
  }
  if (getenv("XPS") && mode == 0) {  // layer 1 on the 128 x 128 tile of the 8 192-row shards
    for (int rep = 0; rep < 2; ++rep) {
      run_dma<0, 0, 2, 1>(g, M, N, K, false); puts("  ^ 128x128 layer-1 LDS-DMA kernel, full");
      run_dma<0, 3, 2, 1>(g, M, N, K, false); puts("  ^ no global stores in the epilogue");
      run_dma<0, 4, 2, 1>(g, M, N, K, false); puts("  ^ one k-step only: prologue + epilogue + stores");
      run_dma<0, 1, 2, 1>(g, M, N, K, false); puts("  ^ no fills in the loop");
    }
    return 0;
  }
  if (getenv("XP") && mode == 0) {
    run_dma<0, 0>(g, M, N, K, false);
    printf("  ^ layer-1 LDS-DMA kernel, full\n");
    run_dma<0, 3>(g, M, N, K, false);
    printf("  ^ no global stores in the epilogue\n");
    run_dma<0, 4>(g, M, N, K, false);
    printf("  ^ one k-step only: epilogue + stores\n");
    run_dma<0, 1>(g, M, N, K, false);
    printf("  ^ no fills in the loop\n");
    return 0;
  }
  if (getenv("XP") && mode == 1) {
    run_dma<1, 5>(g, M, N, K, false); check(N / 256);
    puts("  ^ no ping-pong: free-running waves, one barrier per k-block");
    run_dma<1, 0>(g, M, N, K, false);
    printf("  ^ LDS-DMA kernel, full\n");
    run_dma<1, 1>(g, M, N, K, false);
    printf("  ^ LDS-DMA kernel, no fills in the loop\n");
    run_dma<1, 2>(g, M, N, K, false);
    printf("  ^ LDS-DMA kernel, no fills, no fragment reads: barriers + MFMA\n");
    return 0;
  }
  if (mode == 1) run_dma<1>(g, M, N, K, false); else run_dma<0>(g, M, N, K, false);
  check(N / kDmaBN);
  V(2, 2, 32, 3, 1); check(N / 128);
  V(2, 2, 64, 2, 1); check(N / 128);
  V(2, 2, 32, 3, 2); check(N / 128);
  V(2, 2, 32, 2, 2); check(N / 128);
  V(1, 1, 64, 4, 1); check(N / 64);
  V(1, 1, 32, 6, 2); check(N / 64);
  V(1, 1, 64, 4, 2); check(N / 64);
  quiet = true;
  const char* names[8] = {"128x128x32 w3 pf1", "128x128x64 w2 pf1", "128x128x32 w3 pf2", "128x128x32 w2 pf2", "64x64x64 w4 pf1", "64x64x32 w6 pf2", "64x64x64 w4 pf2", "LDS-DMA 256x256x32"};
  std::vector<std::vector<double>> t(8);
  for (int r = 0; r < 5; ++r) {
    V(2, 2, 32, 3, 1); t[0].push_back(g_us);
    V(2, 2, 64, 2, 1); t[1].push_back(g_us);
    V(2, 2, 32, 3, 2); t[2].push_back(g_us);
    V(2, 2, 32, 2, 2); t[3].push_back(g_us);
    V(1, 1, 64, 4, 1); t[4].push_back(g_us);
    V(1, 1, 32, 6, 2); t[5].push_back(g_us);
    V(1, 1, 64, 4, 2); t[6].push_back(g_us);
    if (mode == 1) run_dma<1>(g, M, N, K, true); else run_dma<0>(g, M, N, K, true);
    t[7].push_back(g_us);
  }
  for (int i = 0; i < 8; ++i) {
    std::sort(t[i].begin(), t[i].end());
    const double med = t[i][t[i].size() / 2], tf = 2.0 * M * N * K / (med * 1e-6) / 1e12;
    printf("%-18s median %8.1f us (min %8.1f max %8.1f)  %6.1f TF algorithmic, %.3f of the fp16 peak executed\n", names[i], med,
           t[i].front(), t[i].back(), tf, 3 * tf / 2516.6);
  }
  return 0;
}
