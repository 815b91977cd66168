// EXPERIMENT (round 2), not part of the product library.  Measured at 32 768 rows (tools/gemm_f16_bench.hip PANEL=1): 8 waves of
// 32 x 64: 52.5 us; 4 waves of 64 x 64: 60.3 us; the 256 x 256 tile kernel: 52.5 us.  Ablation of the 8-wave variant: without any
// epilogue 49.1 us, without MFMAs 35.9 us, fills + fragment reads + barriers alone 19.8 us -- the interleaved epilogue is nearly free
// (3 us), but this k-loop (0.67 LDS reads per MFMA, nothing de-phasing the two waves of a SIMD) runs at half the tile kernel's rate.
// What was kept from it: the direct-store epilogue (activation fragment first, DPP lane swap, buffer stores), now in the tile kernel.
// Layer 1 of the fp16-split discriminator forward for shards of >= ~24 K rows and a SHORT reduction (K <= 192: the K = 2
// workloads, K D = 166 / 162): H = relu(Xs W1^T / (s_x s_w) + b1) as the two fp16 planes of s_h H in block layout.  The same
// arithmetic, in the same order, as every other f16 kernel (three v_mfma_f32_32x32x16_f16 per k-step into one fp32 accumulator
// -- x1 w0, x0 w1, x0 w0 --, k ascending, relu_split4's epilogue), so the hidden layer is bit-identical to theirs.
//
// What bounded the tile kernel (disc_gemm_f16_dma_kernel<0>: 54-56 us per 32 768 rows): its two phases run one after the
// other on every CU at once -- ~14 us of k-loop (no stores anywhere on the chip), then ~14 us in which 256 KiB per CU leave
// through the transposing LDS slabs at the chip's 6.5 TB/s store rate (no MFMA anywhere) -- twice per launch.  The MFMAs
// alone are ~24 us per launch at the rate the power-limited matrix pipes sustain, the stores alone ~21 us; they only have
// to overlap.  This kernel makes every wave do both all the time:
//   * ROW PANEL per workgroup: 128 rows x ALL N columns, walked in chunks of 128 columns.  8 waves (two per SIMD) as 4 (rows)
//     x 2 (columns); per chunk a wave owns 32 rows x 64 columns = 2 accumulator blocks.  (A 4-wave variant with 64 x 64 per
//     wave needs 176 + 128 + 32 registers for fragments and accumulators alone; hipcc spilled ~70 of them to scratch and
//     the reloads' vmcnt(0) waits serialised the kernel on its own stores: 108 us.)
//   * the wave's ACTIVATION fragments (32 rows x K, both planes: 88 registers at K = 176) are loaded ONCE, straight from
//     the block-layout rows of the scaled input in global memory into registers (no de-interleave: the planes are separate), and reused for all N / 128 chunks: the scaled input is read
//     from HBM exactly once, never staged in LDS, and the k-loop's LDS traffic is the weight fragments only (4 reads per
//     6 MFMAs: 32 KiB per k-step for the CU).
//   * the WEIGHTS stream through an 8-stage LDS ring of 16-KiB units (one k-block of one chunk's 128 weight rows, full-line
//     LDS-DMA pieces, the usual chunk swizzle), seven units ahead of the MFMAs; ONE barrier per unit, issued after the
//     first MFMA of the unit's last k-step, publishes the next unit and frees this one's stage.
//   * TWO accumulator sets: while chunk c accumulates, the epilogue of chunk c - 1 (bias, ReLU, plane split, store) is
//     issued in slices between its MFMAs -- ~6 VALU operations and half a store per MFMA -- so the matrix pipe never waits
//     for the epilogue and the stores leave as a steady stream instead of a burst.
//   * the MFMA takes the activation fragment FIRST, so an accumulator register holds one output ROW and the 32 lanes of a
//     lane half 32 consecutive COLUMNS: no transpose through LDS.  Adjacent lanes swap their (p0, p1) words (one DPP move +
//     one v_perm_b32) so that even lanes hold the p0 halves of two columns and odd lanes the p1 halves, and ONE
//     global_store_dword per (row, k-block) writes a full 128-B line of the block layout per lane half.
//   * vmcnt: fills and stores of a wave share the counter.  The wait in front of a unit's barrier is vmcnt(number of
//     YOUNGER fill pieces): with at most that many operations outstanding, at most that many loads are, and loads complete
//     in order among themselves -- so the pieces of the next unit have landed whatever the stores are doing (stores can
//     only make the wait longer, never wrong).
// Rows past M re-read the last row and are stored into the workspace's row padding (amp_disc_workspace_bytes rounds the
// hidden layer up to 128 rows), so there is no store mask in the loop.
#pragma once
#include <type_traits>

#include "disc_gemm_f16_dma.hpp"

namespace amp {

constexpr int kPanelBM = 128, kPanelCH = 128, kPanelStages = 8;
template <int TM>
constexpr int panel_threads() { return 64 * (4 / TM) * 2; }  // TM = 2: 4 waves of 64 x 64 per chunk; TM = 1: 8 waves of 32 x 64
constexpr int kPanelUnit = kPanelCH * 128;                    // bytes of a ring unit: 128 weight rows x one k-block
constexpr int kPanelMaxN = 2048;                              // columns whose scaled bias fits the LDS tail
constexpr int kPanelLdsBytes = kPanelStages * kPanelUnit + kPanelMaxN * 4;  // 128 KiB ring + the scaled bias

__device__ __forceinline__ void panel_wait_fills(int younger_pieces) {  // vmcnt(younger fill pieces), a multiple of 2
  switch (younger_pieces >> 1) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  }
}

// KS = k-steps (of 16) that hold data: ceil(K / 16) <= 12
template <int KS, int TM>
__global__ __launch_bounds__(panel_threads<TM>(), 1) void disc_gemm_f16_panel_kernel(GemmF16Args g) {
  static_assert(KS >= 2 && KS <= 12, "the activation fragments of the whole reduction live in registers");
  // TM = 2 (4 waves of 64 x 64 per chunk) was removed in round 3: slower (60 vs 52 us), 443 registers with the spills parked
  // in AGPRs, and an earlier build of it that spilled to scratch faulted at 65 536 rows (profiles/r02_gemm_f16_l1_panel_experiment.txt)
  static_assert(TM == 1, "only the 8-wave variant (32 x 64 per wave and chunk) is kept");
  constexpr int NKB = (KS + 1) / 2, TN = 2, NST = kPanelStages, kThreads = panel_threads<TM>();
  constexpr int PPW = 16 / (kThreads / 64);  // fill pieces per wave and unit
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 4 row bands of 32 rows x 2 column halves of a chunk
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * kPanelBM;
  if (m0 >= g.M) return;
  const int n_chunks = g.N / kPanelCH;
  const int n_units = n_chunks * NKB;

  // ---- weight ring: unit u = (chunk c, k-block kb) -> stage u % NST; the wave issues pieces 2 wave, 2 wave + 1 of the 16
  const _Float16* wsrc[PPW];
#pragma unroll
  for (int p = 0; p < PPW; ++p) {
    const int r = 8 * (PPW * wave + p) + (lane >> 3);  // weight row inside the chunk
    wsrc[p] = g.W + (int64_t)r * (2 * (int64_t)g.Kp) + 8 * ((lane & 7) ^ ((r >> 1) & 7));
  }
  const int64_t chunk_pitch = (int64_t)kPanelCH * 2 * g.Kp;  // halves between chunks of weight rows
  auto fill_unit = [&](int u) {
    const int c = u / NKB, kb = u - c * NKB;
    unsigned char* dst = lds + (u % NST) * kPanelUnit + (PPW * wave) * 1024;
#pragma unroll
    for (int p = 0; p < PPW; ++p)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[p] + c * chunk_pitch + kb * 64), (lptr_t)(dst + p * 1024), 16, 0, 0);
  };
  // weight fragment addresses inside a stage: row r = 64 wn + 32 b + li, chunk (4 plane + 2 (s & 1) + lh) ^ ((r >> 1) & 7)
  const int swz = (li >> 1) & 7;
  const int wrow = (wn * 64 + li) * 128;
  int cw[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    cw[s][0] = ((2 * s + lh) ^ swz) * 16;
    cw[s][1] = ((4 + 2 * s + lh) ^ swz) * 16;
  }
  struct WFrags { h8 w0[TN], w1[TN]; };
  auto read_w = [&](WFrags& f, int u, int s) {
    const unsigned char* sb = lds + (u % NST) * kPanelUnit + wrow;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      f.w0[b] = *reinterpret_cast<const h8*>(sb + b * 32 * 128 + cw[s & 1][0]);
      f.w1[b] = *reinterpret_cast<const h8*>(sb + b * 32 * 128 + cw[s & 1][1]);
    }
  };

  // ---- prologue: the first NST units in flight (every stage), then the wave's activation fragments (read once, kept in
  // registers) and the scaled bias of all N columns (LDS tail)
  for (int u = 0; u < NST && u < n_units; ++u) fill_unit(u);
  h8 x0[TM][KS], x1[TM][KS];
  {
    const int64_t last = g.M - 1;  // rows past M re-read the last row; their outputs land in the workspace's row padding
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int64_t r = m0 + wm * (32 * TM) + a * 32 + li;
      const _Float16* row = g.A + (r < last ? r : last) * (2 * g.lda);  // block layout: k-block kb = [p0 x 32 | p1 x 32]
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const _Float16* blk = row + (s >> 1) * 64 + (s & 1) * 16 + 8 * lh;  // the lane half's 8 values of k-step s, plane 0
        x0[a][s] = *reinterpret_cast<const h8*>(blk);
        x1[a][s] = *reinterpret_cast<const h8*>(blk + 32);
      }
    }
  }
  // epilogue constants
  const LayerScales sc = layer_scales(g.range, g.amax, g.layer);
  const float s_h = sc.s_out, ds = sc.descale * s_h;
  float* const s_bias = reinterpret_cast<float*>(lds + kPanelStages * kPanelUnit);  // bias * s_h, all N columns
  for (int n = tid; n < g.N; n += kThreads) s_bias[n] = g.bias[n] * s_h;
  // even lanes store the p0 words of columns (li, li + 1), odd lanes the p1 words of (li - 1, li): byte (li >> 1) * 4 of the
  // 64-B plane half; a block's two plane halves are one 128-B line
  const uint32_t sel = (li & 1) ? 0x03020706u : 0x05040100u;  // v_perm_b32(neighbour, own, sel): odd [nb.hi, own.hi], even [own.lo, nb.lo]
  const uint32_t row_pitch = (uint32_t)(2 * g.ldh * (int64_t)sizeof(_Float16));  // bytes between rows of the hidden layer
  // stores: buffer_store_dword with the wave's 32 TM output rows as the buffer (base + range in four SGPRs), a SCALAR offset per
  // item (register r's row, the k-block of (chunk, wn, b)) and ONE per-lane offset (the lane half's 4 rows, the column word):
  // no vector address arithmetic in the loop, and rows outside the wave's band cannot be written
  const uint32_t lane_off = (uint32_t)(4 * lh) * row_pitch + (uint32_t)((li & 1) * 64 + (li >> 1) * 4);
  const int wm_u = __builtin_amdgcn_readfirstlane(wm), wn_u = __builtin_amdgcn_readfirstlane(wn);
  unsigned char* const hwave = reinterpret_cast<unsigned char*>(g.H) + (m0 + wm_u * (32 * TM)) * (int64_t)row_pitch + wn_u * 256;
  const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(hwave, 0, (int)(32u * TM * row_pitch), 0x00020000);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first units and the activation fragments have landed
  __builtin_amdgcn_s_barrier();                      // ... for every wave; the bias table is complete

  fx16 accA[TM][TN], accB[TM][TN];
  WFrags f0, f1;
  read_w(f0, 0, 0);

  // epilogue item (a, r, b) of a finished accumulator set: output row 32 a + (r & 3) + 8 (r >> 2) + 4 lh, k-block
  // (128 c + 64 wn + 32 b) / 32 of the hidden layer: relu_split4's arithmetic (7 VALU operations), the lane swap (2) and one
  // 4-B store per lane.  bs0 / bs1: the lane's scaled bias in blocks 0 / 1 of the chunk.
  auto epi_item = [&](const fx16 (&acc)[TM][TN], const int c, const int idx, const float bs0, const float bs1) {
    const int a = idx >> 5, r = (idx >> 1) & 15, b = idx & 1;  // idx < 32 TM
    const float v = fmaxf(__builtin_fmaf(acc[a][b][r], ds, b ? bs1 : bs0), 0.0f);
    const _Float16 p0 = (_Float16)v;
    const _Float16 p1 = (_Float16)__builtin_fmaf((float)p0, -1.0f, v);
    const uint32_t own = (uint32_t)__builtin_bit_cast(uint16_t, p0) | ((uint32_t)__builtin_bit_cast(uint16_t, p1) << 16);
    const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]: swap adjacent lanes
    const uint32_t word = __builtin_amdgcn_perm(nb, own, sel);
    const int soff = (int)((uint32_t)(a * 32 + (r & 3) + 8 * (r >> 2)) * row_pitch) + c * (kPanelCH * 4) + b * 128;  // wave-uniform
    __builtin_amdgcn_raw_buffer_store_b32(word, hrsrc, (int)lane_off, soff, 0);
  };

  // one chunk: KS k-steps of 6 MFMAs on `cur`; between them the 32 epilogue items of `prev` (chunk c - 1).  FLIP: which
  // fragment set holds k-step 0 (k-step s is in set (s + FLIP) & 1): with an odd KS the sets swap roles from chunk to chunk.
  auto chunk_pass = [&](const int c, fx16 (&cur)[TM][TN], const fx16 (&prev)[TM][TN], auto have_prev_t, auto flip_t) {
    constexpr bool HAVE_PREV = decltype(have_prev_t)::value;
    constexpr int FLIP = decltype(flip_t)::value ? 1 : 0;
    float bs0 = 0.0f, bs1 = 0.0f;
    if (HAVE_PREV) {
      bs0 = s_bias[(c - 1) * kPanelCH + wn * 64 + li];
      bs1 = s_bias[(c - 1) * kPanelCH + wn * 64 + 32 + li];
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[a][b][r] = 0.0f;
    constexpr int kPer = 3 * TM * TN, kItems = 32 * TM;
    constexpr int kSlots = KS * (kPer - 1);  // MFMA slots that carry epilogue work (every MFMA but the first of a k-step)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int u = c * NKB + (s >> 1);
      const bool last_of_unit = (s & 1) == 1 || s == KS - 1;
      WFrags& f = ((s + FLIP) & 1) ? f1 : f0;
      WFrags& fn = ((s + FLIP) & 1) ? f0 : f1;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            const int i = p * (TM * TN) + a * TN + b;  // MFMA index inside the k-step: a constant after unrolling
            const h8 x = p == 0 ? x1[a][s] : x0[a][s];
            const h8 w = p == 1 ? f.w1[b] : f.w0[b];
            cur[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, cur[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) {
              // behind the k-step's first MFMA: the unit hand-over (if this is the unit's last k-step) and the weight
              // fragments of the next k-step
              if (last_of_unit && u + 1 < n_units) {
                const int rest = n_units - 2 - u;  // units after u + 1 ...
                panel_wait_fills(PPW * (rest < NST - 2 ? rest : NST - 2));  // ... of which at most NST - 2 are in flight
                __builtin_amdgcn_s_barrier();
                if (u + NST < n_units) fill_unit(u + NST);  // into the stage unit u is leaving
              }
              if (s + 1 < KS) read_w(fn, u + ((s & 1) ? 1 : 0), s + 1);
              else if (c + 1 < n_chunks) read_w(fn, u + 1, 0);
            } else if (HAVE_PREV) {
              // epilogue slice of this slot: items [64 n / kSlots, 64 (n + 1) / kSlots), n = the slot's number in the chunk
              const int n = s * (kPer - 1) + (i - 1);
#pragma unroll
              for (int item = (kItems * n) / kSlots; item < (kItems * (n + 1)) / kSlots; ++item) epi_item(prev, c - 1, item, bs0, bs1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
    }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  using OddFlip = std::integral_constant<bool, (KS & 1) == 1>;  // odd chunks start on the other fragment set when KS is odd
  chunk_pass(0, accA, accB, No{}, No{});
  int c = 1;
  for (; c + 1 < n_chunks; c += 2) {  // (odd, even) pairs: the accumulator roles are static inside the loop
    chunk_pass(c, accB, accA, Yes{}, OddFlip{});
    chunk_pass(c + 1, accA, accB, Yes{}, No{});
  }
  if (c < n_chunks) chunk_pass(c, accB, accA, Yes{}, OddFlip{});
  // the last chunk's epilogue has nothing to hide behind
  {
    const int c = n_chunks - 1;
    const float bs0 = s_bias[c * kPanelCH + wn * 64 + li], bs1 = s_bias[c * kPanelCH + wn * 64 + 32 + li];
#pragma unroll
    for (int idx = 0; idx < 32 * TM; ++idx) {
      if (c & 1) epi_item(accB, c, idx, bs0, bs1);
      else epi_item(accA, c, idx, bs0, bs1);
    }
  }
}

}  // namespace amp
