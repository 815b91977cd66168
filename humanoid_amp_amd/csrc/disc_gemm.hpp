// fp32-MFMA GEMM building block of the discriminator (included by disc.hip and tools/gemm_bench.hip).
//   MODE 0: C = relu(A W^T + bias) stored [M, N]
//   MODE 1: partial[m, nt] = sum over the tile's columns of relu(A W^T + bias)[m, n] * w3[n]
//   MODE 2: C = (A W^T) [* (mask > 0)] [+ C]   plain product for the training step's backward GEMMs
// A [M, lda] and W [N, Kp] are row-major fp32 with 16-B aligned rows, Kp % BK == 0.
//
// Tile: BM x BN x BK_, 4 waves as 2 x 2, each wave (BM/2) x (BN/2) = TM x TN accumulators of
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).  LDS rows are padded by 4 floats so the ds_read_b128 operand
// fetches are conflict-free; lane (li, lh) consumes k = (BK_/2)*lh + s at MFMA step s (a fixed k-permutation shared
// by A and B -- legal because fp32 addition order inside a dot product is not part of the contract).
// STAGES_ = 1: one LDS stage, two barriers per k-tile.  STAGES_ = 2: two stages, one barrier per k-tile.
// Global -> LDS staging is register-prefetched one k-tile ahead; the issue order is pinned with sched_barrier because
// hipcc otherwise sinks the prefetch loads below the MFMAs and exposes their whole latency.
#pragma once
#include "amp_common.hpp"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));  // native vector: HIP's float4 struct turns into memcpy -> scratch

namespace amp {

struct GemmArgs {
  const float* A; int64_t lda; int64_t M; int32_t K;
  const float* W; int32_t Kp;
  const float* bias; int32_t N;
  float* C; int64_t ldc;                               // mode 0 / 2 output
  const float* mask; int64_t ldmask; int32_t accumulate; // mode 2: optional elementwise gate (mask > 0) and C +=
  int32_t k_slices; int64_t slice_stride;              // mode 2: split-K, slice s writes its partial product to C + s * slice_stride
  const float* w3; float* partial; int32_t n_tiles;    // mode 1 output [M, n_tiles]
  int32_t m_tiles;
};

// XCD-aware renumbering: hardware deals workgroups round-robin over the 8 XCDs; give each XCD a contiguous
// run of (row tile, column tile) pairs with the column tile fastest (speed only, never correctness).
__device__ __forceinline__ bool tile_of_block(const GemmArgs& g, int& mt, int& nt, int* slice = nullptr) {
  const int tiles = g.m_tiles * g.n_tiles;
  const int total = tiles * (g.k_slices > 1 ? g.k_slices : 1);
  const int per_xcd = (total + 7) / 8;
  int v = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (v >= total) return false;
  if (slice) *slice = v / tiles;
  v %= tiles;
  mt = v / g.n_tiles;
  nt = v - mt * g.n_tiles;
  return true;
}

// Workgroup barrier that waits for this wave's LDS traffic only (__syncthreads() would also drain vmcnt, i.e. stall
// on the global prefetch that is meant to stay in flight).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int BM_, int BN_, int BK_>
struct Stage {
  static constexpr int LDT = BK_ + 4, QK = BK_ / 4, RPP = kBlock / QK;  // row pitch, 16-B pieces per row, rows per pass
  f4 a[BM_ / RPP], b[BN_ / RPP];
  __device__ __forceinline__ void load(const GemmArgs& g, int64_t m0, int n0, int kt, int tid) {
    const int c4 = tid % QK, r4 = tid / QK;
    const float* w = g.W + (int64_t)(n0 + r4) * g.Kp + kt * BK_ + 4 * c4;
    const int64_t step = (int64_t)RPP * g.Kp;
#pragma unroll
    for (int i = 0; i < BN_ / RPP; ++i) b[i] = *reinterpret_cast<const f4*>(w + i * step);
    const int64_t last = g.M - 1;  // rows past M re-read the last row; their results are never stored
    const float* base = g.A + kt * BK_ + 4 * c4;
#pragma unroll
    for (int i = 0; i < BM_ / RPP; ++i) {
      const int64_t m = m0 + r4 + RPP * i;
      a[i] = *reinterpret_cast<const f4*>(base + (m < last ? m : last) * g.lda);
    }
  }
  __device__ __forceinline__ void store(float* As, float* Bs, int tid) const {
    const int c4 = tid % QK, r4 = tid / QK;
#pragma unroll
    for (int i = 0; i < BN_ / RPP; ++i) *reinterpret_cast<f4*>(&Bs[(r4 + RPP * i) * LDT + 4 * c4]) = b[i];
#pragma unroll
    for (int i = 0; i < BM_ / RPP; ++i) *reinterpret_cast<f4*>(&As[(r4 + RPP * i) * LDT + 4 * c4]) = a[i];
  }
};

// SWAP = false: acc[a][b] = X_a W_b^T  (C/D rows = activation rows, lanes = weight rows / output columns)
// SWAP = true : acc[a][b] = (W_b X_a^T) (C/D rows = output columns n, lanes = activation rows m): the transposed tile,
//               which turns a reduction over output columns into a per-lane sum over accumulator registers.
template <int TM, int TN, int BK_, bool SWAP>
__device__ __forceinline__ void compute_tile(const float* As, const float* Bs, int arow, int brow, int lh,
                                             floatx16 (&acc)[TM][TN]) {
  constexpr int LDT = BK_ + 4, NQ = BK_ / 8;  // 16-B pieces per lane per operand row
  const f4* pa = reinterpret_cast<const f4*>(&As[arow * LDT + (BK_ / 2) * lh]);
  const f4* pb = reinterpret_cast<const f4*>(&Bs[brow * LDT + (BK_ / 2) * lh]);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    f4 x[TM], y[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) x[a] = pa[a * (32 * LDT / 4) + q];
#pragma unroll
    for (int b = 0; b < TN; ++b) y[b] = pb[b * (32 * LDT / 4) + q];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(y[b][c], x[a][c], acc[a][b], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x2f32(x[a][c], y[b][c], acc[a][b], 0, 0, 0);
  }
}

template <int BM_, int BN_, int BK_, int STAGES_, int MODE, int MINW_>
__global__ __launch_bounds__(kBlock, MINW_) void disc_gemm_kernel(GemmArgs g) {
  constexpr int TM = BM_ / 64, TN = BN_ / 64, LDT = BK_ + 4;
  constexpr int STAGE_F = (BM_ + BN_) * LDT;
  constexpr int EP_F = 4 * 32 * (TN * 32 + 4);  // epilogue transpose region (mode 0)
  constexpr int SMEM_F = STAGES_ * STAGE_F > EP_F ? STAGES_ * STAGE_F : EP_F;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_F];
  int mt, nt, slice = 0;
  if (!tile_of_block(g, mt, nt, &slice)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)mt * BM_;
  const int n0 = nt * BN_;
  // k-tile range of this workgroup (split-K: slice s owns a contiguous run of k-tiles)
  const int nk_all = g.Kp / BK_;
  const int per_slice = (MODE == 2 && g.k_slices > 1) ? (nk_all + g.k_slices - 1) / g.k_slices : nk_all;
  const int k0 = slice * per_slice;
  const int nk = k0 + per_slice < nk_all ? k0 + per_slice : nk_all;  // one past the last k-tile
  if (MODE == 2 && g.k_slices > 1) g.C += (int64_t)slice * g.slice_stride;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int arow = wm * (TM * 32) + li, brow = wn * (TN * 32) + li;
  Stage<BM_, BN_, BK_> stg;
  if (k0 < nk) {
    stg.load(g, m0, n0, k0, tid);
    stg.store(smem, smem + BM_ * LDT, tid);
  }
  if (STAGES_ == 2) {
    if (k0 + 1 < nk) stg.load(g, m0, n0, k0 + 1, tid);
    __syncthreads();
    for (int kt = k0; kt < nk; ++kt) {
      float* cur = smem + ((kt - k0) & 1) * STAGE_F;
      float* nxt = smem + ((kt - k0 + 1) & 1) * STAGE_F;
      if (kt + 1 < nk) stg.store(nxt, nxt + BM_ * LDT, tid);   // k-tile kt+1 (in registers since kt-1)
      if (kt + 2 < nk) stg.load(g, m0, n0, kt + 2, tid);        // in flight under this tile's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      compute_tile<TM, TN, BK_, MODE == 1>(cur, cur + BM_ * LDT, arow, brow, lh, acc);
      lds_barrier();
    }
  } else {
    __syncthreads();
    for (int kt = k0; kt < nk; ++kt) {
      if (kt + 1 < nk) stg.load(g, m0, n0, kt + 1, tid);        // in flight under this tile's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      compute_tile<TM, TN, BK_, MODE == 1>(smem, smem + BM_ * LDT, arow, brow, lh, acc);
      lds_barrier();
      if (kt + 1 < nk) {
        stg.store(smem, smem + BM_ * LDT, tid);
        lds_barrier();
      }
    }
  }
  __syncthreads();

  // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) --------
  if (MODE == 0 || MODE == 2) {
    // bias + ReLU (mode 0) / optional gate + accumulate (mode 2), then transpose each wave's 32 x (TN*32) slab through its own LDS region (the staging tiles are
    // dead) so that a lane stores 16 B and consecutive lanes cover contiguous bytes of one output row
    constexpr int W = TN * 32, EPL = W + 4, QPR = W / 4;  // row width, padded row, 16-B pieces per row
    float* ep = smem + wave * (32 * EPL);
    float bias[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) bias[b] = MODE == 0 ? g.bias[n0 + wn * W + b * 32 + li] : 0.0f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
        for (int b = 0; b < TN; ++b)
          ep[row * EPL + b * 32 + li] = MODE == 0 ? fmaxf(acc[a][b][r] + bias[b], 0.0f) : acc[a][b][r];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < (32 * QPR) / 64; ++i) {
        const int idx = lane + 64 * i, row = idx / QPR, q = idx % QPR;
        f4 v = *reinterpret_cast<const f4*>(&ep[row * EPL + 4 * q]);
        const int64_t grow = m0 + wm * (TM * 32) + a * 32 + row;
        if (grow < g.M) {
          const int col = n0 + wn * W + 4 * q;
          if (MODE == 2) {
            if (g.mask) {
              const f4 mk = *reinterpret_cast<const f4*>(&g.mask[grow * g.ldmask + col]);
#pragma unroll
              for (int c = 0; c < 4; ++c) v[c] = mk[c] > 0.0f ? v[c] : 0.0f;
            }
            if (g.accumulate) v += *reinterpret_cast<const f4*>(&g.C[grow * g.ldc + col]);
          }
          *reinterpret_cast<f4*>(&g.C[grow * g.ldc + col]) = v;
        }
      }
      __syncthreads();
    }
  } else {
    // The accumulators hold the TRANSPOSED tile (rows = output columns n, lanes = activation rows m), so
    // relu(acc + b2) . w3 over this wave's TN*32 columns is a per-lane sum over registers: register r of lane half lh
    // is column (r & 3) + 8 * (r >> 2) + 4 * lh, i.e. four consecutive columns per group of four registers.
    float* red = smem;  // [2][BM]
    const f4* bias4 = reinterpret_cast<const f4*>(g.bias + n0 + wn * (TN * 32) + 4 * lh);
    const f4* w34 = reinterpret_cast<const f4*>(g.w3 + n0 + wn * (TN * 32) + 4 * lh);
    float sum[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) sum[a] = 0.0f;
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        const f4 bs = bias4[b * 8 + grp * 2], ws = w34[b * 8 + grp * 2];  // columns b*32 + 8*grp + 4*lh + [0, 4)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int i = 0; i < 4; ++i) sum[a] += fmaxf(acc[a][b][4 * grp + i] + bs[i], 0.0f) * ws[i];
      }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const float v = sum[a] + __shfl_xor(sum[a], 32, 64);  // the other lane half holds the other columns
      if (lh == 0) red[wn * BM_ + wm * (TM * 32) + a * 32 + li] = v;
    }
    __syncthreads();
    if (tid < BM_) {
      const int64_t row = m0 + tid;
      if (row < g.M) g.partial[row * g.n_tiles + nt] = red[tid] + red[BM_ + tid];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// "TT" product for the training step's weight gradients:  C[m][n] (+)= sum_k A[k][m] * W[k][n]  -- both operands with the
// REDUCTION index as their row (the batch: dW = dY^T X with dY [rows, m], X [rows, n] exactly as the forward / backward
// kernels leave them).  The NT kernel above needs both operands k-contiguous, which cost the step eight explicit
// transposes (dH2^T, H1^T, dH1^T, Xs^T, a1^T, dg^T, a2^T, e1^T: 10 launches, ~94 us and ~400 MB of traffic per step).
// Here a k-tile is BK_ rows of BM_ / BN_ contiguous floats (fully coalesced loads, stored to LDS as they come); the MFMA's
// A / B operand of lane (li, lh) at step s is the single float [2 s + lh][row0 + li]: 32 consecutive floats per lane half,
// conflict-free 4-B LDS reads, one per MFMA.  Rows k >= g.K read as zero (ragged batch).  g.Kp is W's row pitch here.
// Split-K and the epilogue as MODE 2 (no mask).
template <int BM_, int BN_, int BK_, int MINW_>
__global__ __launch_bounds__(kBlock, MINW_) void disc_gemm_tt_kernel(GemmArgs g) {
  constexpr int TM = BM_ / 64, TN = BN_ / 64, LDA = BM_ + 4, LDB = BN_ + 4;
  constexpr int STAGE_F = BK_ * (LDA + LDB);
  constexpr int EP_F = 4 * 32 * (TN * 32 + 4);
  constexpr int SMEM_F = STAGE_F > EP_F ? STAGE_F : EP_F;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_F];
  int mt, nt, slice = 0;
  if (!tile_of_block(g, mt, nt, &slice)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)mt * BM_;
  const int n0 = nt * BN_;
  const int nk_all = (g.K + BK_ - 1) / BK_;
  const int per_slice = g.k_slices > 1 ? (nk_all + g.k_slices - 1) / g.k_slices : nk_all;
  const int k0 = slice * per_slice;
  const int nk = k0 + per_slice < nk_all ? k0 + per_slice : nk_all;
  if (g.k_slices > 1) g.C += (int64_t)slice * g.slice_stride;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  constexpr int QA = BM_ / 4, QB = BN_ / 4, RA = kBlock / QA, RB = kBlock / QB;  // 16-B pieces per row, rows per pass
  constexpr int NA = BK_ / RA, NB = BK_ / RB;
  static_assert(BK_ % RA == 0 && BK_ % RB == 0, "the k-tile is a whole number of load passes");
  f4 ra[NA], rb[NB];
  float* const As = smem;
  float* const Bs = smem + BK_ * LDA;
  auto load = [&](const int kt) {
    const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int64_t k = (int64_t)kt * BK_ + tid / QA + RA * i;
      ra[i] = k < g.K ? *reinterpret_cast<const f4*>(g.A + k * g.lda + m0 + 4 * (tid % QA)) : zero;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int64_t k = (int64_t)kt * BK_ + tid / QB + RB * i;
      rb[i] = k < g.K ? *reinterpret_cast<const f4*>(g.W + k * (int64_t)g.Kp + n0 + 4 * (tid % QB)) : zero;
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f4*>(&As[(tid / QA + RA * i) * LDA + 4 * (tid % QA)]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f4*>(&Bs[(tid / QB + RB * i) * LDB + 4 * (tid % QB)]) = rb[i];
  };
  const int arow = wm * (TM * 32) + li, brow = wn * (TN * 32) + li;
  if (k0 < nk) {
    load(k0);
    store();
  }
  __syncthreads();
  for (int kt = k0; kt < nk; ++kt) {
    if (kt + 1 < nk) load(kt + 1);  // in flight under this tile's MFMAs
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < BK_ / 2; ++s2) {
      float x[TM], y[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) x[a] = As[(2 * s2 + lh) * LDA + arow + a * 32];
#pragma unroll
      for (int b = 0; b < TN; ++b) y[b] = Bs[(2 * s2 + lh) * LDB + brow + b * 32];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[a], y[b], acc[a][b], 0, 0, 0);
    }
    lds_barrier();
    if (kt + 1 < nk) {
      store();
      lds_barrier();
    }
  }
  __syncthreads();
  // epilogue as MODE 2 (no mask): C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  constexpr int W = TN * 32, EPL = W + 4, QPR = W / 4;
  float* ep = smem + wave * (32 * EPL);
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int b = 0; b < TN; ++b) ep[row * EPL + b * 32 + li] = acc[a][b][r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (32 * QPR) / 64; ++i) {
      const int idx = lane + 64 * i, row = idx / QPR, q = idx % QPR;
      f4 v = *reinterpret_cast<const f4*>(&ep[row * EPL + 4 * q]);
      const int64_t grow = m0 + wm * (TM * 32) + a * 32 + row;
      if (grow < g.M) {
        const int col = n0 + wn * W + 4 * q;
        if (g.accumulate) v += *reinterpret_cast<const f4*>(&g.C[grow * g.ldc + col]);
        *reinterpret_cast<f4*>(&g.C[grow * g.ldc + col]) = v;
      }
    }
    __syncthreads();
  }
}

}  // namespace amp
