// EXPERIMENT (round 2), not part of the product library: measured within 5 % of the 8-wave 128 x 128 LDS-DMA tile at every
// shard size (29.7 vs 29.4 us at 8 192 rows; deeper activation rings 5 + 3 / 6 + 3 were SLOWER) -- a 128 x 128 tile is bound
// by the ~68 GB/s a CU takes in through LDS-DMA (1 MiB of fills per tile), not by its wave layout.  Kept for tools/gemm_f16_bench.hip.
// GEMMs of the fp16-split discriminator forward for SMALL shards (a few thousand rows: the 8 192-env shards of the
// multi-GPU configurations): the same arithmetic, in the same order, as disc_gemm_f16_kernel / disc_gemm_f16_dma_kernel
// (three v_mfma_f32_32x32x16_f16 per k-step into one fp32 accumulator, k ascending, transposed accumulator tile), so the
// results are bit-identical to theirs whatever kernel a shard size selects.
//
// Why another tile.  A shard of 8 192 rows is 64 tiles of 256 x 256 -- a quarter of the chip.  The 8-wave kernel with
// 128 x 128 tiles gives every CU one tile but its waves are 64 x 32: six MFMAs (192 matrix-pipe cycles) per k-step
// against six 1-KiB fragment reads, i.e. the eight waves ask the LDS for 48 KiB per k-step = 384 cycles at 128 B / clk,
// exactly the 2 x 192 MFMA cycles a SIMD has to cover them with: the tile ran at 0.35 of the pipe (29 us).  Here the
// 128 x 128 tile is computed by FOUR waves of 64 x 64: twelve MFMAs (384 cycles) per k-step against eight reads, 32 KiB
// = 256 LDS cycles per k-step for the workgroup, one wave per SIMD owning its matrix pipe.
//
//   * operands in block layout, filled by full-line LDS-DMA pieces (8 rows x 128 B) exactly like the large-shard kernel;
//     a k-block (32 values) of one operand = 128 rows x 128 B = 16 KiB; chunk swizzle c ^ ((r >> 1) & 7).
//   * what bounded every 128 x 128 variant at ~29 us per 8 192-row launch (8-wave ping-pong, 8-wave free-running, this
//     tile with 3 / 4 symmetric stages: all within 5 %) is the fill LATENCY: the activations of a small shard come from
//     the Infinity Cache (33 MB of hidden layer do not fit the 4-MB L2s), ~1-2 us away, and a CU that keeps one or two
//     k-blocks (32-64 KiB) in flight moves 1 MiB per tile at ~36 GB/s.  So the two operands get SEPARATE rings: NA stages
//     for the activations (far away: NA - 1 k-blocks in flight, filled by waves 0-1) and NW stages for the weights
//     (L2-resident: NW - 1 in flight, filled by waves 2-3); 5 + 3 stages = 128 KiB, 6 + 3 = 144 KiB.
//   * the waves run free: ONE barrier per k-block, placed BETWEEN the two k-steps' MFMA groups.  Loop invariant: the
//     fragments of (k-block q, k-step 0) are in registers.  Then: first MFMA of k-step 0 | read the fragments of k-step 1
//     | the other 11 MFMAs | counted vmcnt wait for this wave's pieces of k-block q + 1, barrier | refill the stages
//     k-block q - 1 used (every wave is past its reads: they fed MFMAs issued before the barrier) | first MFMA of k-step 1
//     | read the fragments of (q + 1, k-step 0) | the other 11 MFMAs.  Every fragment read is issued eleven MFMAs (352
//     cycles) before its first use.  A refill target must not be a stage still being read, hence >= 3 stages per ring.
//   * LDS-DMA visibility (MI355X_MICROARCH.md): the ISSUING wave's vmcnt wait, then a barrier every reader passes
//     before its first read of the stage.
#pragma once
#include "disc_gemm_f16_dma.hpp"

namespace amp {

constexpr int kDma4Threads = 256, kDma4BM = 128, kDma4BN = 128;
template <int NA, int NW>
constexpr int dma4_lds_bytes() { return (NA + NW) * 128 * 128; }

// `younger` k-blocks of 8 pieces each may still be in flight behind the one being waited for (fills retire in issue order)
__device__ __forceinline__ void dma4_wait_pieces(int younger) {
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
}

template <int MODE, int NA, int NW>
__global__ __launch_bounds__(kDma4Threads, 1) void disc_gemm_f16_dma4_kernel(GemmF16Args g) {
  static_assert(NA >= 3 && NW >= 3 && NA <= 8 && NW <= 8, "three to eight stages per ring");
  constexpr int BM = kDma4BM, TM = 2, TN = 2, kOp = BM * 128;  // bytes of one operand's k-block
  constexpr int kRingW = NA * kOp;                                // the weight ring follows the activation ring
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  int mt, nt;
  if (!f16_tile_of_block(g, mt, nt)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)mt * BM;
  const int n0 = nt * kDma4BN;
  const int nq = g.Kp / kDmaKB;

  // ---- fill plan: waves 0-1 fill activation rows [64 (wave & 1), + 64), waves 2-3 the weight rows likewise; piece j =
  // rows 8 j .. 8 j + 7 of that half; lane l: row + (l >> 3), stored chunk (l & 7) = source chunk (l & 7) ^ ((row >> 1) & 7)
  const _Float16* src[8];
  {
    const int64_t last = g.M - 1;  // rows past M re-read the last row; their results are never stored
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = (wave & 1) * 64 + j * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      if (wave < 2) {
        const int64_t m = m0 + r < last ? m0 + r : last;
        src[j] = g.A + m * (2 * g.lda) + 8 * c;
      } else {
        src[j] = g.W + (int64_t)(n0 + r) * (2 * (int64_t)g.Kp) + 8 * c;
      }
    }
  }
  const bool fills_a = wave < 2;                 // wave-uniform: this wave's ring
  const int ns = fills_a ? NA : NW;              // its depth
  const int ring_base = (fills_a ? 0 : kRingW) + (wave & 1) * (8 * 1024);
  auto fill = [&](int q) {  // the wave's 8 pieces of k-block q -> stage q % ns of its ring
    unsigned char* sb = lds + ring_base + (q % ns) * kOp;
#pragma unroll
    for (int j = 0; j < 8; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + q * 64), (lptr_t)(sb + j * 1024), 16, 0, 0);
  };

  // ---- fragment addresses (bytes inside a stage), as in disc_gemm_f16_dma_kernel
  const int swz = (li >> 1) & 7;
  const int arow = (wm * 64 + li) * 128, brow = (wn * 64 + li) * 128;  // inside a stage of the respective ring
  int ca[2][2], cb[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    ca[s][0] = ((MODE == 1 ? 2 * s + lh : 4 * s + 2 * lh) ^ swz) * 16;
    ca[s][1] = ((MODE == 1 ? 4 + 2 * s + lh : 4 * s + 2 * lh + 1) ^ swz) * 16;
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

  struct Frags { h8 x0[TM], x1[TM], w0[TN], w1[TN]; };
  auto read_frags = [&](Frags& f, const int q, const int s) {
    const unsigned char* sb = lds + (q % NA) * kOp;            // activation stage of k-block q
    const unsigned char* sw = lds + kRingW + (q % NW) * kOp;   // weight stage
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      if (MODE == 1) {
        f.x0[a] = *reinterpret_cast<const h8*>(sb + arow + a * 32 * 128 + ca[s][0]);
        f.x1[a] = *reinterpret_cast<const h8*>(sb + arow + a * 32 * 128 + ca[s][1]);
      } else {
        const uv4 lo = *reinterpret_cast<const uv4*>(sb + arow + a * 32 * 128 + ca[s][0]);
        const uv4 hi = *reinterpret_cast<const uv4*>(sb + arow + a * 32 * 128 + ca[s][1]);
        uv4 q0, q1;
        q0[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x05040100u); q1[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x07060302u);
        q0[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x05040100u); q1[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302u);
        q0[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x05040100u); q1[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x07060302u);
        q0[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x05040100u); q1[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302u);
        f.x0[a] = __builtin_bit_cast(h8, q0);
        f.x1[a] = __builtin_bit_cast(h8, q1);
      }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      f.w0[b] = *reinterpret_cast<const h8*>(sw + brow + b * 32 * 128 + cb[s][0]);
      f.w1[b] = *reinterpret_cast<const h8*>(sw + brow + b * 32 * 128 + cb[s][1]);
    }
  };
  // the 12 MFMAs of a k-step in the product order of every other f16 kernel (w0 x1, w1 x0, w0 x0); `first`: only the
  // very first one / all but the first (the fragment reads of the NEXT k-step are issued between the two, see below)
  auto mfmas = [&](const Frags& f, const bool first) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
        if ((a == 0 && b == 0) == first) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.w0[b], f.x1[a], acc[a][b], 0, 0, 0);
    if (first) return;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.w1[b], f.x0[a], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.w0[b], f.x0[a], acc[a][b], 0, 0, 0);
  };

  // ---- prologue: k-blocks 0 .. ns - 2 of the wave's ring in flight; k-block 0 visible to everyone
  for (int s = 0; s < ns - 1 && s < nq; ++s) fill(s);
  dma4_wait_pieces(nq - 1 < ns - 2 ? nq - 1 : ns - 2);
  __builtin_amdgcn_s_barrier();
  // hipcc's waitcnt pass merges the loop's back edge conservatively: the wait in front of a k-step's first MFMA is a
  // full lgkmcnt(0).  So the reads of the NEXT k-step's fragments are issued right AFTER that first MFMA (under its 32
  // matrix-pipe cycles): nothing younger is outstanding at the wait, and every read still has eleven MFMAs (352 cycles)
  // to land before it is needed.
  Frags f0, f1;
  read_frags(f0, 0, 0);
  for (int q = 0; q < nq; ++q) {
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f0, true);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(f1, q, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f0, false);
    __builtin_amdgcn_sched_barrier(0);
    if (q + 1 < nq) {
      // this wave's pieces of k-block q + 1 have landed; issued so far: k-blocks .. q + ns - 2
      const int rest = nq - 2 - q;  // k-blocks after q + 1
      dma4_wait_pieces(rest < ns - 3 ? rest : ns - 3);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // the stage k-block q - 1 used: its last reads (k-step 1) fed MFMAs every wave issued before arriving at the barrier
      if (q + ns - 1 < nq) fill(q + ns - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfmas(f1, true);
    __builtin_amdgcn_sched_barrier(0);
    if (q + 1 < nq) read_frags(f0, q + 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f1, false);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();  // every wave is done with the stages: the scratch below reuses them

  // ---- epilogue: register r of lane half lh is output column (r & 3) + 8 (r >> 2) + 4 lh of the 32-wide block,
  //      lane li is activation row li
  const LayerScales sc = layer_scales(g.range, g.amax, g.layer);
  const float descale = sc.descale;
  const fv4* bias4 = reinterpret_cast<const fv4*>(g.bias + n0 + wn * (32 * TN) + 4 * lh);
  if (MODE == 0) {
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
        constexpr int CPR = 8 * TN;  // chunks per row in memory order: k-block b, plane pl, quarter qq
        const int idx = lane + 64 * i, row = idx / CPR, ch = idx % CPR, b = ch >> 3, pl = (ch >> 2) & 1, qq = ch & 3;
        const h8 v = *reinterpret_cast<const h8*>(&ep[pl * 32 * EPL + row * EPL + b * 32 + 8 * qq]);
        const int64_t grow = m0 + wm * (32 * TM) + a * 32 + row;
        if (grow < g.M) *reinterpret_cast<h8*>(&g.H[grow * (2 * g.ldh) + (int64_t)((n0 + wn * CW) >> 5) * 64 + 8 * ch]) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    return;
  }
  // canonical partial logits, one per (row, 32-column block): see disc_gemm_f16_kernel's MODE 1 epilogue
  const fv4* w34 = reinterpret_cast<const fv4*>(g.w3 + n0 + wn * (32 * TN) + 4 * lh);
  float* red = reinterpret_cast<float*>(lds);  // [2 TN][BM]
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
  constexpr int BPT = 2 * TN;  // 32-column blocks per tile
  const int n_blocks = g.N >> 5;
  for (int e = tid; e < BM * BPT; e += kDma4Threads) {
    const int r = e / BPT, j = e - r * BPT;
    const int64_t row = m0 + r;
    if (row < g.M) g.partial[row * n_blocks + nt * BPT + j] = red[j * BM + r];
  }
}

}  // namespace amp
