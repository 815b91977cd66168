// Large-shard LAYER-2 GEMM of the fp16-split discriminator forward, FOUR waves per workgroup: the same arithmetic, in the same
// order, as the other f16 kernels (three v_mfma_f32_32x32x16_f16 per k-step into one fp32 accumulator -- w0 x1, w1 x0,
// w0 x0 --, k ascending, transposed accumulator tile, canonical partial logits), so the results are bit-identical to theirs.
//
// Why.  The 8-wave 256 x 256 ping-pong kernel (disc_gemm_f16_dma.hpp) ends at 0.51 of the fp16 pipe: its ablation with
// neither fills nor fragment reads (barriers + MFMAs only) takes 79 of the 80 us of a 32 768-row layer-2 launch -- two wave
// groups hand the matrix pipe to each other through four barriers per k-block, and every hand-over costs ~500 cycles next
// to 768 cycles of MFMAs.  Here ONE wave per SIMD owns its matrix pipe for the whole tile:
//   * 256 x 256 tile, 4 waves as 2 x 2, each 128 x 128 = 16 accumulator blocks (256 accumulator registers: the AGPR half
//     of the 512-register file a lone wave per SIMD has), 48 MFMAs (1 536 pipe cycles) per k-step against 16 fragment
//     reads: 64 KiB of LDS reads per k-step for the CU instead of 96 (a third of the LDS bandwidth).
//   * operands in block layout, full-line LDS-DMA pieces (8 rows x 128 B), chunk swizzle c ^ ((r >> 1) & 7) -- as before.
//     A stage = one k-block (32 values) of both operands = 64 KiB, TWO stages; a wave issues 16 pieces per k-block
//     (waves 0-1 the activation rows, waves 2-3 the weight rows).
//   * ONE barrier per k-block, in the middle of it.  Invariant at the top of iteration q: the fragments of (q, k-step 0)
//     are in registers.  First half: the 48 MFMAs of k-step 0, with the 16 reads of (q, k-step 1) issued one after each of
//     the first MFMAs.  Then: lgkmcnt(0) (every read of stage q & 1 by this wave is complete), vmcnt(0) (its pieces of
//     k-block q + 1, issued a whole k-block ago, have landed), BARRIER -- which therefore both publishes k-block q + 1 and
//     frees stage q & 1.  Second half: the 48 MFMAs of k-step 1, with the 16 pieces of k-block q + 2 (-> stage q & 1) and
//     the 16 reads of (q + 1, k-step 0) issued between them.  No read, fill or barrier ever waits in front of an idle
//     matrix pipe except the barrier's own skew; a fill has a full k-block (~3 000 cycles) to land.
//   * hipcc's waitcnt pass puts a full lgkmcnt(0) in front of a half's first MFMA (loop back edge); the reads of a half are
//     issued AFTER that MFMA, so nothing young is outstanding when it waits.
//
// EXPERIMENT (round 2), not part of the product library.  Measured (tools/gemm_f16_bench.hip W4=1, profiles/r02_gemm_f16_w4.txt):
// 76.8 us per 32 768-row launch against 78.5 us for the 8-wave kernel in the microbenchmark -- and 75.2 us with its fills,
// fragment reads AND barrier compiled out: the bare MFMA stream is the bound.  On operands that change from one MFMA to the
// next the matrix pipes sustain 0.55-0.63 of the nominal 2 516.8 TFLOP/s (0.70-0.93 on constant operands; ENTROPY=1 in the
// microbench, profiles/r02_mfma_entropy_ceiling.txt; amp_calibrate_mfma_f16 in the library) -- the chip is power-limited.
// INSIDE the step, where the hidden layer arrives from the Infinity Cache right behind layer 1, this kernel's single
// k-block of fill look-ahead made it slower than the 8-wave kernel (85 vs 80 us), so the product keeps the latter.
#pragma once
#include <type_traits>

#include "disc_gemm_f16_dma.hpp"

namespace amp {

constexpr int kW4Threads = 256, kW4BM = 256, kW4BN = 256;
constexpr int kW4LdsBytes = 2 * (kW4BM + kW4BN) * 128;  // 128 KiB: two stages of both operands

__global__ __launch_bounds__(kW4Threads, 1) void disc_gemm_f16_w4_kernel(GemmF16Args g) {
  constexpr int BM = kW4BM, TM = 4, TN = 4, kOp = BM * 128, kStage = 2 * kOp, NP = 16;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  int mt, nt;
  if (!f16_tile_of_block(g, mt, nt)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)mt * BM;
  const int n0 = nt * kW4BN;
  const int nq = g.Kp / kDmaKB;

  // ---- fill plan: waves 0-1 fill the activation rows [128 (wave & 1), + 128), waves 2-3 the weight rows likewise; piece j =
  // rows 8 j .. 8 j + 7 of that half; lane l: row + (l >> 3), stored chunk (l & 7) = source chunk (l & 7) ^ ((row >> 1) & 7)
  const _Float16* src[NP];
  {
    const int64_t last = g.M - 1;  // rows past M re-read the last row; their results are never stored
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = (wave & 1) * 128 + j * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      if (wave < 2) {
        const int64_t m = m0 + r < last ? m0 + r : last;
        src[j] = g.A + m * (2 * g.lda) + 8 * c;
      } else {
        src[j] = g.W + (int64_t)(n0 + r) * (2 * (int64_t)g.Kp) + 8 * c;
      }
    }
  }
  const int fill_base = (wave >> 1) * kOp + (wave & 1) * (NP * 1024);
  auto piece = [&](int q, int j) {  // piece j of the wave's share of k-block q -> stage q & 1
    __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + q * 64), (lptr_t)(lds + (q & 1) * kStage + fill_base + j * 1024), 16, 0, 0);
  };

  // ---- fragment addresses (bytes inside a stage), as in disc_gemm_f16_dma_kernel
  const int swz = (li >> 1) & 7;
  const int arow = (wm * 128 + li) * 128, brow = kOp + (wn * 128 + li) * 128;
  int ca[2][2], cb[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    ca[s][0] = ((2 * s + lh) ^ swz) * 16;      // plane 0, k-step s, lane half lh
    ca[s][1] = ((4 + 2 * s + lh) ^ swz) * 16;  // plane 1
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

  // A fragment set = 16 reads of 16 B per lane: both planes of the wave's 4 activation blocks and 4 weight blocks
  struct Frags { h8 x0[TM], x1[TM], w0[TN], w1[TN]; };
  auto read_one = [&](Frags& f, const unsigned char* sb, const int s, const int i) {  // read i of 16
    if (i < 2 * TM) {
      const int a = i >> 1;
      if ((i & 1) == 0) f.x0[a] = *reinterpret_cast<const h8*>(sb + arow + a * 32 * 128 + ca[s][0]);
      else f.x1[a] = *reinterpret_cast<const h8*>(sb + arow + a * 32 * 128 + ca[s][1]);
    } else {
      const int b = (i - 2 * TM) >> 1;
      if ((i & 1) == 0) f.w0[b] = *reinterpret_cast<const h8*>(sb + brow + b * 32 * 128 + cb[s][0]);
      else f.w1[b] = *reinterpret_cast<const h8*>(sb + brow + b * 32 * 128 + cb[s][1]);
    }
  };
  // One half of a k-block: the 48 MFMAs of fragment set `f` in the product order of every f16 kernel (per accumulator:
  // w0 x1, then w1 x0, then w0 x0); `between(i)` runs right after MFMA i (i = 0 .. 47) and is where the reads and fill
  // pieces of the next half are issued, pinned in place by sched_barrier.
  auto half = [&](const Frags& f, auto&& between) {
    int i = 0;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const h8 w = p == 1 ? f.w1[b] : f.w0[b];
          const h8 x = p == 0 ? f.x1[a] : f.x0[a];
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, acc[a][b], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          between(i);
          __builtin_amdgcn_sched_barrier(0);
          ++i;
        }
  };

  // ---- prologue: k-blocks 0 and 1 in flight; k-block 0 visible to everyone; fragments of (0, k-step 0)
#pragma unroll
  for (int j = 0; j < NP; ++j) piece(0, j);
  if (nq > 1) {
#pragma unroll
    for (int j = 0; j < NP; ++j) piece(1, j);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  Frags f0, f1;
#pragma unroll
  for (int i = 0; i < 16; ++i) read_one(f0, lds, 0, i);

  // one k-block; MORE: a k-block q + 1 exists (barrier + its k-step-0 reads), REFILL: a k-block q + 2 exists (its pieces).
  // Compile-time flags: run-time conditions inside the unrolled halves would put 32 scalar branches between the MFMAs.
  auto kblock = [&](const int q, auto more_t, auto refill_t) {
    constexpr bool MORE = decltype(more_t)::value, REFILL = decltype(refill_t)::value;
    const unsigned char* sb = lds + (q & 1) * kStage;
    const unsigned char* sn = lds + ((q + 1) & 1) * kStage;
    // first half: k-step 0; the reads of k-step 1 ride behind the first 16 MFMAs
    half(f0, [&](int i) {
      if (i < 16) read_one(f1, sb, 1, i);
    });
    if (MORE) {
      // every read of stage q & 1 by this wave is complete; its pieces of k-block q + 1 have landed
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    // second half: k-step 1; behind its MFMAs: the reads of (q + 1, k-step 0), then the pieces of k-block q + 2 (-> the
    // stage this k-block is leaving)
    half(f1, [&](int i) {
      if (i < 16) {
        if (MORE) read_one(f0, sn, 0, i);
      } else if (i < 16 + 2 * NP && (i & 1) == 0) {
        if (REFILL) piece(q + 2, (i - 16) >> 1);
      }
    });
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  int q = 0;
  for (; q + 2 < nq; ++q) kblock(q, Yes{}, Yes{});
  if (q + 1 < nq) kblock(q++, Yes{}, No{});
  kblock(q, No{}, No{});
  __syncthreads();  // every wave is done with the stages: the scratch below reuses them

  // ---- epilogue: register r of lane half lh is output column (r & 3) + 8 (r >> 2) + 4 lh of the 32-wide block,
  //      lane li is activation row li
  const LayerScales sc = layer_scales(g.range, g.amax, g.layer);
  const float descale = sc.descale;
  const fv4* bias4 = reinterpret_cast<const fv4*>(g.bias + n0 + wn * (32 * TN) + 4 * lh);
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
  for (int e = tid; e < BM * BPT; e += kW4Threads) {
    const int r = e / BPT, j = e - r * BPT;
    const int64_t row = m0 + r;
    if (row < g.M) g.partial[row * n_blocks + nt * BPT + j] = red[j * BM + r];
  }
}

}  // namespace amp
