// ABLATION COPY of csrc/disc_gemm_f16_dma.hpp's kernel for tools/gemm_f16_bench.hip -- NOT part of the product library.
// XP != 0 variants produce wrong results on purpose (1 = no fills in the loop, 2 = also no fragment reads, 3 = MODE 0
// without the epilogue's global stores, 4 = MODE 0 epilogue only) or a different schedule (5 = free-running waves, one
// barrier per k-block: correct results).  XP = 0 is the product schedule.
#pragma once
#include "disc_gemm_f16_dma.hpp"  // product kernel, DmaTile, split_rows_blocks_kernel, pointer typedefs

namespace amp {

// Args: A = activations in block layout, row pitch 2 * lda halves;
// W = weights in block layout, row pitch 2 * Kp halves; MODE 0 output H in block layout, row pitch 2 * ldh halves.
// XP != 0: ablations for tools/gemm_f16_bench.hip (wrong results): 1 = no fills in the loop, 2 = also no fragment reads,
// 3 = MODE 0 without the global stores of the epilogue, 4 = MODE 0 epilogue only (one k-block), 5 = no ping-pong (free-running
// waves, one barrier per k-block; correct results: 106 us against 80 us at 32 768 rows of layer 2)
template <int MODE, int XP = 0, int TM = 4, int TN = 2>
__global__ __launch_bounds__(kDmaThreads, (DmaTile<TM, TN>::kWgPerCu)) void disc_gemm_f16_dma_xp_kernel(GemmF16Args g) {
  using T = DmaTile<TM, TN>;
  constexpr int BM = T::BM, BN = T::BN, kOpA = T::kA, kStage = T::kStage;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  int mt, nt;
  if (!f16_tile_of_block(g, mt, nt)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int grp = wave >> 2;
  const int64_t m0 = (int64_t)mt * BM;
  const int n0 = nt * BN;
  const int nq = XP == 4 ? 1 : g.Kp / kDmaKB;  // k-blocks

  // ---- fill plan: a piece is 8 rows x 128 B.  Group 0's wave w fills a quarter of the activation rows, group 1's
  // wave 4 + w a quarter of the weight rows: NP pieces each.  lane l: row 8 j + (l >> 3) of the wave's quarter, stored
  // chunk (l & 7) = source chunk (l & 7) ^ ((row >> 1) & 7)
  constexpr int NPA = BM / 32, NPB = BN / 32, NP = NPA > NPB ? NPA : NPB;
  const int np = grp == 0 ? NPA : NPB;
  const _Float16* src[NP];
  {
    const int64_t last = g.M - 1;  // rows past M re-read the last row; their results are never stored
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = (wave & 3) * (8 * np) + j * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      if (grp == 0) {
        const int64_t m = m0 + r < last ? m0 + r : last;
        src[j] = g.A + m * (2 * g.lda) + 8 * c;
      } else {
        src[j] = g.W + (int64_t)(n0 + r) * (2 * (int64_t)g.Kp) + 8 * c;
      }
    }
  }
  const int fill_base = grp * kOpA + (wave & 3) * (np * 1024);
  auto fill = [&](int q, int stage) {  // the wave's pieces of k-block q
    unsigned char* sb = lds + stage * kStage + fill_base;
#pragma unroll
    for (int j = 0; j < NP; ++j)
      if (j < np) __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + q * 64), (lptr_t)(sb + j * 1024), 16, 0, 0);
  };
  // outstanding pieces after a k-block's fill, for the counted wait of the prologue
  (void)np;

  // ---- fragment addresses (bytes inside a stage): row r, chunk c -> r * 128 + (c ^ ((r >> 1) & 7)) * 16; the wave's
  // rows start at multiples of 32, so (r >> 1) & 7 = (li >> 1) & 7.  Block layout: plane pl, k-step s, lane half lh ->
  // chunk 4 pl + 2 s + lh.  Pair layout (MODE 0 activations): the lane's eight values are chunks 4 s + 2 lh, + 1.
  const int swz = (li >> 1) & 7;
  const int arow = (wm * (32 * TM) + li) * 128, brow = kOpA + (wn * (32 * TN) + li) * 128;
  int ca[2][2], cb[2][2];  // [k-step][plane] byte offsets inside a row
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    ca[s][0] = ((2 * s + lh) ^ swz) * 16;
    ca[s][1] = ((4 + 2 * s + lh) ^ swz) * 16;
    cb[s][0] = ((2 * s + lh) ^ swz) * 16;
    cb[s][1] = ((4 + 2 * s + lh) ^ swz) * 16;
  }

  fx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  h8 x0[TM], x1[TM], w0[TN], w1[TN];
  auto read_frags = [&](const unsigned char* sb, const int s) {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      x0[a] = *reinterpret_cast<const h8*>(sb + arow + a * 32 * 128 + ca[s][0]);
      x1[a] = *reinterpret_cast<const h8*>(sb + arow + a * 32 * 128 + ca[s][1]);
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      w0[b] = *reinterpret_cast<const h8*>(sb + brow + b * 32 * 128 + cb[s][0]);
      w1[b] = *reinterpret_cast<const h8*>(sb + brow + b * 32 * 128 + cb[s][1]);
    }
  };
  // matrix segment: the 24 MFMAs at raised priority between two barriers; `drain`: the wave's outstanding pieces must
  // have landed before the closing barrier (group 0, second k-step of a k-block)
  auto matrix_segment = [&](bool drain) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[b], x1[a], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[b], x0[a], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[b], x0[a], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  fill(0, 0);
  if (nq > 1) {
    fill(1, 1);
    // k-block 0 has landed, k-block 1 (np pieces) is in flight
    if (np == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (np == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (np == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();  // k-block 0 is visible to every wave
  if (XP == 5) {
    // ablation / experiment: no ping-pong -- every wave runs reads -> MFMAs -> reads -> MFMAs freely, ONE barrier per k-block
    auto mfmas = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[b], x1[a], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[b], x0[a], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[b], x0[a], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int q = 0; q < nq; ++q) {
      const unsigned char* sb = lds + (q & 1) * kStage;
      if (q >= 1 && q + 1 < nq) fill(q + 1, (q + 1) & 1);  // its last reads were retired in front of the barrier below
      read_frags(sb, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mfmas();
      read_frags(sb, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mfmas();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  if (grp == 1) __builtin_amdgcn_s_barrier();
  for (int q = 0; q < nq; ++q) {
    const unsigned char* sb = lds + (q & 1) * kStage;
    // R0: fragments of k-step 0; the other stage (k-block q - 1: its last reads were retired in front of a barrier
    // this wave has passed) takes k-block q + 1 (k-block 1 was issued in the prologue)
    if (XP < 2 || q == 0) read_frags(sb, 0);
    if (XP == 0 && q >= 1 && q + 1 < nq) fill(q + 1, (q + 1) & 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    matrix_segment(false);
    // R1: fragments of k-step 1; group 1's pieces of k-block q + 1 must have landed before the next barrier
    if (XP < 2 || q == 0) read_frags(sb, 1);
    if (grp == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    matrix_segment(grp == 0);  // group 0's pieces: behind its MFMAs
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  }
  __syncthreads();  // every wave is done with the stages: the scratch below reuses them

  // ---- epilogue: register r of lane half lh is output column (r & 3) + 8 (r >> 2) + 4 lh of the 32-wide block,
  //      lane li is activation row li
  const LayerScales sc = layer_scales(g.range, g.amax, g.layer);
  const float descale = sc.descale;
  const fv4* bias4 = reinterpret_cast<const fv4*>(g.bias + n0 + wn * (32 * TN) + 4 * lh);
  if (MODE == 0) {
    // relu(. + bias) -> the two planes of s_h H in block layout, transposed through a wave-private LDS slab per 32
    // rows so that a lane stores 16 B and sixteen lanes cover the 256 contiguous bytes a row gets from this wave
    // (two k-blocks x [p0 | p1])
    constexpr int CW = 32 * TN, EPL = CW + 8;  // the wave's columns, padded slab row (halves)
    const float s_h = sc.s_out;
    _Float16* ep = reinterpret_cast<_Float16*>(lds) + wave * (2 * 32 * EPL);
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int grp4 = 0; grp4 < 4; ++grp4) {
          const fv4 bs = bias4[b * 8 + grp4 * 2] * s_h;
          fv4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[a][b][4 * grp4 + i];
          h4 p0, p1;
          relu_split4(v, descale * s_h, bs, p0, p1);
          const int col = b * 32 + 8 * grp4 + 4 * lh;
          *reinterpret_cast<h4*>(&ep[li * EPL + col]) = p0;
          *reinterpret_cast<h4*>(&ep[32 * EPL + li * EPL + col]) = p1;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 4 * TN; ++i) {
        // 8 TN chunks per row in memory order: k-block b, plane pl, quarter qq
        constexpr int CPR = 8 * TN;
        const int idx = lane + 64 * i, row = idx / CPR, ch = idx % CPR, b = ch >> 3, pl = (ch >> 2) & 1, qq = ch & 3;
        const h8 v = *reinterpret_cast<const h8*>(&ep[pl * 32 * EPL + row * EPL + b * 32 + 8 * qq]);
        const int64_t grow = m0 + wm * (32 * TM) + a * 32 + row;
        if (grow < g.M && (XP != 3 || v[0] == (_Float16)12345.0f))
          *reinterpret_cast<h8*>(&g.H[grow * (2 * g.ldh) + (int64_t)((n0 + wn * CW) >> 5) * 64 + 8 * ch]) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    return;
  }
  const fv4* w34 = reinterpret_cast<const fv4*>(g.w3 + n0 + wn * (32 * TN) + 4 * lh);
  // canonical partial logits, one per (row, 32-column block): see disc_gemm_f16_kernel's MODE 1 epilogue
  float* red = reinterpret_cast<float*>(lds);  // [4 TN][BM]
  float sum[TM][TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
#pragma unroll
    for (int a = 0; a < TM; ++a) sum[a][b] = 0.0f;
#pragma unroll
    for (int grp4 = 0; grp4 < 4; ++grp4) {
      const fv4 bs = bias4[b * 8 + grp4 * 2], ws = w34[b * 8 + grp4 * 2];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[a][b] += fmaxf(acc[a][b][4 * grp4 + i] * descale + bs[i], 0.0f) * ws[i];
    }
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const float v = sum[a][b] + __shfl_xor(sum[a][b], 32, 64);  // the other lane half holds the other columns
      if (lh == 0) red[(wn * TN + b) * BM + wm * (32 * TM) + a * 32 + li] = v;
    }
  __syncthreads();
  constexpr int BPT = 4 * TN;  // 32-column blocks per tile
  const int n_blocks = g.N >> 5;
  for (int e = tid; e < BM * BPT; e += kDmaThreads) {
    const int r = e / BPT, j = e - r * BPT;
    const int64_t row = m0 + r;
    if (row < g.M) g.partial[row * n_blocks + nt * BPT + j] = red[j * BM + r];
  }
}

}  // namespace amp
