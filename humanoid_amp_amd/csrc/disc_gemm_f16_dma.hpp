// GEMMs of the fp16-split discriminator forward for large shards: the same arithmetic, in the same order, as
// disc_gemm_f16_kernel (three v_mfma_f32_32x32x16_f16 per k-step into one fp32 accumulator, transposed accumulator
// tile; MODE 0: bias + ReLU -> fp16 planes of the hidden layer (its activations are the scaled input); MODE 1: bias +
// ReLU + dot(w3) -> partial logits), rebuilt around what bounds them on MI355X -- the operand fill path:
//
//   * measured with a fill-only probe (tools/gemm_f16_bench.hip, FILL=1: the kernel's DMA pattern, no compute): LDS-DMA
//     pieces that take 32 B from each row move 6.9 TB/s chip-wide, 64 B 11.9 TB/s, a whole 128-B line 17.1 TB/s.  A
//     256 x 256 tile needs 1.07 GB of fills per 65 536-row launch: 154 us with 32-B row segments (= the whole kernel),
//     63 us with full lines.
//   * so every operand is stored in BLOCK layout: row r, k-block kb (32 values) = 128 contiguous bytes holding both
//     planes, [p0: 32 halves][p1: 32 halves] (the scaled input is produced in the same layout: 128 B per 32
//     values).  One LDS-DMA piece = 8 rows x 128 B = 1 KiB, full cache lines only.
//   * 256 x 256 workgroup tile, 512 threads = 8 waves as 2 (rows) x 4 (columns), each wave 128 x 64 = 4 x 2
//     accumulator blocks; one workgroup per CU, two waves per SIMD.
//   * stage = 2 k-steps (one k-block) of both operands = 2 x 256 rows x 128 B = 64 KB; two stages (128 KB).
//   * a piece lands lane-linearly (wave-uniform base + 16 B x lane), so rows cannot be padded: the 16-B chunk c of row
//     r sits at chunk c ^ ((r >> 1) & 7) -- applied to the per-lane SOURCE address of the fill and to the fragment
//     reads alike -- which spreads the 16 lanes of a ds_read_b128 group over the 16 slots of a bank row.
//   * every k-step is a LOAD segment (12 ds_read_b128 of fragments; MODE 0: + 32 v_perm_b32) followed by a MATRIX
//     segment (24 MFMAs at s_setprio 1) bracketed by two barriers; the two wave groups (waves 0-3 / 4-7: one wave of
//     each per SIMD) run the sequence one barrier apart, so a SIMD's matrix pipe alternates between its two waves.
//
// Fill schedule (interval = one barrier-to-barrier slot; group 0 runs k-block q in intervals 4q..4q+3 as R0 M0 R1 M1,
// group 1 one interval later).  Stage (q + 1) & 1 holds k-block q - 1, last read in interval 4q - 1 (group 1's R1).
//   group 0 fills the ACTIVATION half (from the Infinity Cache / HBM): its 8 pieces of k-block q + 1 are issued in
//           R0(q) (interval 4q) and awaited (vmcnt(0)) behind the MFMAs of M1(q) (interval 4q + 3);
//   group 1 fills the WEIGHT half (L2-resident): its 8 pieces are issued in its R0(q) (interval 4q + 1) and awaited at
//           the end of its R1(q) (interval 4q + 3).
// Both waits sit in front of the barrier that ends interval 4q + 3, which every reader passes before the first read of
// k-block q + 1 (interval 4q + 4): the LDS-DMA visibility rule of MI355X_MICROARCH.md (counted wait of the ISSUING wave
// + a barrier the reader has passed).  Every refill is issued behind a barrier that follows the lgkmcnt(0) of the
// stage's last reads.  The slow operand has three intervals (>= 2 400 cycles) to land, the L2-warm one two.
#pragma once
#include <type_traits>

#include "disc_gemm_f16.hpp"

namespace amp {

constexpr int kDmaThreads = 512, kDmaBM = 256, kDmaBN = 256;  // the default tile (TM = 4, TN = 2)
constexpr int kDmaKB = 32;                                  // values per k-block = 2 k-steps
constexpr int kDmaLdsBytes = 2 * (kDmaBM + kDmaBN) * 128;   // 128 KB: two stages of both operands
// Tile geometry: 8 waves as 2 (rows) x 4 (columns), each wave (32 TM) x (32 TN); workgroup tile (64 TM) x (128 TN).
template <int TM, int TN, int NS = 2>
struct DmaTile {
  static constexpr int BM = 64 * TM, BN = 128 * TN;
  static constexpr int kA = BM * 128, kB = BN * 128;          // bytes of one operand in a stage
  static constexpr int kStage = kA + kB;
  static constexpr int kLds = NS * kStage;
  static constexpr int kWgPerCu = kLds <= 80 * 1024 ? 2 : 1;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// planes of scale[0] * src[rows, cols] (row pitch ld_src floats) -> block layout dst[rows][kp / 32][2][32] halves;
// columns in [cols, kp) are zero.  One thread per four columns.
// `scale_is_amax`: scale[0] is the operand's abs-max, the scale is plane_scale() of it (training-step operands).
static __global__ __launch_bounds__(kBlock) void split_rows_blocks_kernel(const float* __restrict__ src, int64_t rows, int cols,
                                                                   int64_t ld_src, const float* __restrict__ scale,
                                                                   _Float16* __restrict__ dst, int kp, int scale_is_amax = 0) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x, per_row = kp / 4;
  if (q >= rows * per_row) return;
  const int64_t r = q / per_row;
  const int c = (int)(q - r * per_row) * 4;
  const float s = scale_is_amax ? plane_scale(scale[0]) : scale[0];
  fv4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = c + i < cols ? src[r * ld_src + c + i] * s : 0.0f;
  h4 p0, p1;
  split_planes4(v, p0, p1);
  _Float16* blk = dst + r * (2 * (int64_t)kp) + (c >> 5) * 64 + (c & 31);
  *reinterpret_cast<h4*>(blk) = p0;
  *reinterpret_cast<h4*>(blk + 32) = p1;
}

// Args: A = activations in block layout, row pitch 2 * lda halves (lda values per row);
// W = weights in block layout, row pitch 2 * Kp halves; MODE 0 output H in block layout, row pitch 2 * ldh halves.
// (Ablated variants of this kernel -- no fills, no fragment reads, no stores, free-running waves -- live in
// tools/experiments/disc_gemm_f16_dma_xp.hpp for tools/gemm_f16_bench.hip; the product kernel carries none.)
// KB2 = 3 / 4 (small tiles): a matrix segment holds BOTH k-steps of a k-block -- R M per k-block instead of R0 M0 R1 M1 -- over a
// ring of KB2 stages.  A 128 x 128 tile has 6 MFMAs (192 matrix-pipe cycles) per k-step and wave: with two barriers per k-step
// the barrier-bracketed segments alone cost 20 of the 29 us of an 8 192-row layer-2 launch (tools/gemm_f16_bench.hip XPS=1,
// profiles/r03_gemm_f16_small_tile_ablation.txt: bare MFMA stream 11.7 us).  Twelve MFMAs per segment halve the barriers, and the
// extra stages give the fills the latency cover the two dropped intervals took away (k-block q + KB2 - 1 is issued in R(q); k-block
// q + 1 is awaited before the barrier in front of R(q + 1)'s first read, counted: vmcnt leaves only the younger k-blocks in
// flight).  Measured at 8 192 rows: two stages 31.8 us (no gain over the k-step schedule's 31.5), three 29.1, four 28.2.
// Accumulation order per accumulator is unchanged (k ascending, the same three products per k-step): bit-identical results.
// (KB2 = the number of stages of that ring: 3, or 4 = one more k-block of fill cover.)
#ifndef AMP_L1_STORE_AUX
// Cache-policy bits of the hidden layer's stores: sc1 (agent scope: the line is written through the XCD's L2 instead of
// sitting in it dirty until the next 4 MB of the 134-MB stream push it out).  Measured, same box, 32 768-row launches: aux 0
// 53.1 us, 2 (non-temporal) 53.5, 16 (sc1) 51.7, 17 (sc1 + sc0) 51.9, 18 53.3; inside the step at 65 536 envs layer 1
// 108.0-108.7 -> 105.0-106.5 us (the env launch that follows gives ~2 of the 3 us back: 0.326 -> 0.324 ms per step), at 8 192
// envs 59.2-59.6 -> 57.3-58.3 us per step.
// (The FILLS keep the default policy: non-temporal / sc1 on the activation fills cost layer 2 +8-11 us and layer 1 +4-7 us per
// step, gpurun_out/r03_y.)
#define AMP_L1_STORE_AUX 16
#endif
#ifdef AMP_DMA_TIMELINE  // diagnostic builds only (tools/gemm_f16_bench.hip TIMELINE=1, tools/small_shard_timeline.py): per-workgroup phase stamps
__device__ unsigned long long* g_dma_timeline;  // [rows][8]: start, k-loop start, k-loop end, stores issued / partials reduced, stores done, hw id
// (row = block index, + 1024 for layer-2 launches, so that one buffer takes both launches of a small shard's step)
extern "C" int amp_debug_dma_timeline(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_dma_timeline), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#define AMP_DMA_STAMP(slot)                                                                                           \
  do {                                                                                                                \
    if (g_dma_timeline && threadIdx.x == 0) g_dma_timeline[(size_t)stamp_row * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define AMP_DMA_STAMP(slot) do {} while (0)
#endif

template <int MODE, int TM = 4, int TN = 2, int KB2 = 0>
__global__ __launch_bounds__(kDmaThreads, (DmaTile<TM, TN, KB2 ? KB2 : 2>::kWgPerCu)) void disc_gemm_f16_dma_kernel(GemmF16Args g) {
  using T = DmaTile<TM, TN, KB2 ? KB2 : 2>;
  static_assert(KB2 == 0 || KB2 == 3 || KB2 == 4, "KB2 = stages of the k-block-per-segment ring");
  static_assert(!KB2 || MODE == 1, "the k-block-per-segment schedule is wired for layer 2 (no zero-padding k-step skip, no split-K)");
  constexpr int BM = T::BM, BN = T::BN, kOpA = T::kA, kStage = T::kStage;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
#ifdef AMP_DMA_TIMELINE
  unsigned stamp_row = blockIdx.x + (MODE == 1 ? 1024u : 0u);
#endif
  AMP_DMA_STAMP(0);
  int mt, nt, slice = 0;
  if (MODE == 2 && g.k_slices > 1) {
    // split-K: the grid is k_slices copies of the tile grid; slice s owns the k-blocks [s * nq, (s + 1) * nq)
    const int m_tiles = g.m_tiles;
    g.m_tiles = m_tiles * g.k_slices;
    if (!f16_tile_of_block(g, mt, nt)) return;
    slice = mt / m_tiles;
    mt -= slice * m_tiles;
  } else if (!f16_tile_of_block(g, mt, nt)) {
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: everything derived from it (group, fill duty, tile offsets) stays in SGPRs
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int grp = wave >> 2;
  // Row origin of tile mt.  MODE 0 / 1: the last row tile of a ragged M is moved UP so that it ends at M (its first rows are
  // computed twice, by this tile and by the one above it, from the same operands: the same bits land twice) -- every tile is
  // then a full tile, the per-lane fill offsets below do not depend on the tile, and no fill ever leaves the matrix.  MODE 2
  // (one tile per workgroup, possibly accumulating into C) keeps mt * BM and clamps its rows instead.
  auto tile_origin = [&](const int mt_) -> int64_t {
    const int64_t o = (int64_t)mt_ * BM;
    return (MODE != 2 && o + BM > g.M && g.M >= BM) ? g.M - BM : o;
  };
  int64_t m0 = tile_origin(mt);  // (MODE 0: a persistent workgroup moves on to its next tile, see the end of the epilogue)
  int n0 = nt * BN;
  const int nq = (MODE == 2 && g.k_slices > 1) ? g.Kp / kDmaKB / g.k_slices : g.Kp / kDmaKB;  // k-blocks (of this slice)
  const int q0 = slice * nq;
  const int ksteps = (MODE != 2 && g.ksteps > 0) ? g.ksteps : 2 * nq;

  // ---- fill plan: a piece is 8 rows x 128 B.  Group 0's wave w fills a quarter of the activation rows, group 1's
  // wave 4 + w a quarter of the weight rows: NP pieces each.  lane l: row 8 j + (l >> 3) of the wave's quarter, stored
  // chunk (l & 7) = source chunk (l & 7) ^ ((row >> 1) & 7)
  constexpr int NPA = BM / 32, NPB = BN / 32, NP = NPA > NPB ? NPA : NPB;
  const int np = grp == 0 ? NPA : NPB;
  // A piece's source = a wave-uniform base (the tile's first row of the wave's operand, k-block q0: SGPRs) + a 32-bit
  // per-lane offset (row inside the tile x row pitch + swizzled chunk): eight VGPRs instead of eight 64-bit pointers.
  uint32_t soff[NP];
  const unsigned char* sbase;
  const uint32_t pitch = (uint32_t)(4 * (grp == 0 ? g.lda : (int64_t)g.Kp));  // bytes per row: 2 * ld halves
  {
    // rows past M (only when M < BM, or in MODE 2's last row tile) re-read the last valid row; their results are never stored
    const int64_t left = MODE == 2 ? g.M - m0 : g.M;
    const int rows_ok = grp == 0 ? (left < BM ? (int)left : BM) : BN;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = (wave & 3) * (8 * np) + j * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      const int rr = r < rows_ok ? r : rows_ok - 1;
      soff[j] = (uint32_t)rr * pitch + 16u * (uint32_t)c;
    }
  }
  auto plan_fills = [&]() {  // the wave-uniform base of the tile at (m0, n0)
    const unsigned char* base = grp == 0 ? reinterpret_cast<const unsigned char*>(g.A) + m0 * (int64_t)pitch
                                         : reinterpret_cast<const unsigned char*>(g.W) + (int64_t)n0 * pitch;
    base += (int64_t)q0 * 128;
    // wave-uniform by construction (grp, m0, n0 are): pin it into SGPRs
    const uint64_t bits = reinterpret_cast<uint64_t>(base);
    sbase = reinterpret_cast<const unsigned char*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bits >> 32)) << 32) |
                                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits));
  };
  plan_fills();
  const int fill_base = grp * kOpA + (wave & 3) * (np * 1024);
  auto fill = [&](int q, int stage) {  // the wave's pieces of k-block q
    unsigned char* sb = lds + stage * kStage + fill_base;
    const unsigned char* qb = sbase + q * 128;
#pragma unroll
    for (int j = 0; j < NP; ++j)
      if (j < np) __builtin_amdgcn_global_load_lds((gptr_t)(qb + soff[j]), (lptr_t)(sb + j * 1024), 16, 0, 0);
  };
  // outstanding pieces after a k-block's fill, for the counted wait of the prologue
  (void)np;

  // ---- fragment addresses (bytes inside a stage): row r, chunk c -> r * 128 + (c ^ ((r >> 1) & 7)) * 16; the wave's
  // rows start at multiples of 32, so (r >> 1) & 7 = (li >> 1) & 7.  Block layout: plane pl, k-step s, lane half lh ->
  // chunk 4 pl + 2 s + lh (activations and weights alike).
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
  h8 x0[TM], x1[TM], w0[TN], w1[TN];
  // ---- MODE 1 (layer 2) runs on v_mfma_f32_16x16x32_f16 (disc_gemm_f16.hpp): one MFMA covers a whole k-block.  Fragment of
  // a 16-row block at row R (a multiple of 16), plane pl: lane (i = lane & 15, kq = lane >> 4) reads chunk 4 pl + kq of row
  // R + i, stored at (4 pl + kq) ^ ((i >> 1) & 7) -- the same swizzle; a ds_read_b128's four 16-lane groups still touch
  // every bank once.  A k-block is R0 (all weight fragments + the activation fragments of the wave's upper TM 16-row
  // blocks) M0 (3 x TM x 2 TN MFMAs of 16 cycles = the old 24 of 32) R1 (the lower TM blocks) M1.
  fx4 c16[2 * TM][2 * TN];
  h8 xf0[TM], xf1[TM], wf0[2 * TN], wf1[2 * TN];
  const int i16 = lane & 15, kq = lane >> 4, swz16 = (i16 >> 1) & 7;
  const int arow16 = (wm * (32 * TM) + i16) * 128, brow16 = kOpA + (wn * (32 * TN) + i16) * 128;
  const int ch16[2] = {(kq ^ swz16) * 16, ((4 + kq) ^ swz16) * 16};  // plane 0 / 1
  // MODE 0 (layer 1: lanes of the result = output COLUMNS) feeds the two MFMAs of a 32-column block with the weight rows
  //   rho(beta, i) = 4 (i >> 1) + (i & 1) + 2 beta,  beta = 0 / 1:  {0,1,4,5,...,28,29} and {2,3,6,7,...,30,31},
  // so that after the adjacent-lane exchange of the epilogue a lane holds FOUR consecutive columns of a row (one word from
  // each MFMA): an 8-byte store, a whole 128-B line per row and instruction.  The gather is free: a per-lane row offset;
  // its swizzle term is ((rho >> 1) & 7) = (2 (i >> 1) + beta) & 7, and a ds_read_b128's 16-lane groups still cover the banks.
  int waddr0[2][2];  // [beta][plane]: byte offset of the lane's chunk of its weight row inside a stage
#pragma unroll
  for (int be = 0; be < 2; ++be) {
    const int rho = 4 * (i16 >> 1) + (i16 & 1) + 2 * be, sw = (2 * (i16 >> 1) + be) & 7;
    waddr0[be][0] = kOpA + (wn * (32 * TN) + rho) * 128 + (kq ^ sw) * 16;
    waddr0[be][1] = kOpA + (wn * (32 * TN) + rho) * 128 + ((4 + kq) ^ sw) * 16;
  }
  auto read_w16 = [&](const unsigned char* sb) {
#pragma unroll
    for (int b = 0; b < 2 * TN; ++b) {
      if constexpr (MODE == 0) {
        wf0[b] = *reinterpret_cast<const h8*>(sb + waddr0[b & 1][0] + (b >> 1) * 32 * 128);
        wf1[b] = *reinterpret_cast<const h8*>(sb + waddr0[b & 1][1] + (b >> 1) * 32 * 128);
      } else {
        wf0[b] = *reinterpret_cast<const h8*>(sb + brow16 + b * 16 * 128 + ch16[0]);
        wf1[b] = *reinterpret_cast<const h8*>(sb + brow16 + b * 16 * 128 + ch16[1]);
      }
    }
  };
  auto read_x16 = [&](const unsigned char* sb, const int half) {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      xf0[a] = *reinterpret_cast<const h8*>(sb + arow16 + (half * TM + a) * 16 * 128 + ch16[0]);
      xf1[a] = *reinterpret_cast<const h8*>(sb + arow16 + (half * TM + a) * 16 * 128 + ch16[1]);
    }
  };
  auto mfmas16 = [&](auto half_c) {  // the 3 TM 2 TN MFMAs of one row half, products in the order of every f16 kernel
    constexpr int H = decltype(half_c)::value;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < 2 * TN; ++b)
        c16[H * TM + a][b] = MODE == 1 ? mfma16(wf0[b], xf1[a], c16[H * TM + a][b]) : mfma16(xf1[a], wf0[b], c16[H * TM + a][b]);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < 2 * TN; ++b)
        c16[H * TM + a][b] = MODE == 1 ? mfma16(wf1[b], xf0[a], c16[H * TM + a][b]) : mfma16(xf0[a], wf1[b], c16[H * TM + a][b]);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < 2 * TN; ++b)
        c16[H * TM + a][b] = MODE == 1 ? mfma16(wf0[b], xf0[a], c16[H * TM + a][b]) : mfma16(xf0[a], wf0[b], c16[H * TM + a][b]);
  };
  using half0_t = std::integral_constant<int, 0>;
  using half1_t = std::integral_constant<int, 1>;
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
  auto matrix_segment = [&](bool drain, const int half = 0) {  // half: MODE 1, which TM 16-row blocks the MFMAs are for
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    // the same three products per accumulator in every f16 kernel: (w0, x1), (w1, x0), (w0, x0).  MODE 1 passes the WEIGHT
    // fragment first (accumulator registers = output columns: the 512 -> 1 layer is a per-lane sum); MODE 0 passes the
    // ACTIVATION fragment first (registers = output rows, lanes = columns: rows store without a transpose).  The element
    // arithmetic is the same either way -- the MFMA's reduction order depends on k alone.
    if constexpr (MODE != 2) {
      if (half == 0) mfmas16(half0_t{});
      else mfmas16(half1_t{});
    } else {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc[a][b] = MODE != 1 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(x1[a], w0[b], acc[a][b], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[b], x1[a], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc[a][b] = MODE != 1 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[a], w1[b], acc[a][b], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[b], x0[a], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc[a][b] = MODE != 1 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[a], w0[b], acc[a][b], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[b], x0[a], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  unsigned vb = blockIdx.x;  // the (virtual) block whose tile is being computed
  bool first_tile = true;
  for (;;) {  // MODE 0: one pass per tile of a persistent workgroup; the other modes leave from their epilogues
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
#pragma unroll
  for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
    for (int b = 0; b < 2 * TN; ++b) c16[a][b] = fx4{0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (KB2) {
    h8 y0[2 * TM], y1[2 * TM];  // activation fragments of the wave's 2 TM 16-row blocks, both planes (weights: wf0 / wf1)
    constexpr int NS = KB2;
    auto wait_younger = [&](const int younger) {  // all of the wave's pieces but those of the `younger` newest k-blocks have landed
      const int pieces = younger * np;
      if (pieces <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (pieces == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (pieces == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (pieces == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (pieces == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (pieces == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (3 x ... never occurs with NS <= 4 and np a power of two <= 8)
    };
    for (int q = 0; q < NS - 1 && q < nq; ++q) fill(q, q);
    wait_younger((nq < NS - 1 ? nq : NS - 1) - 1);
    __builtin_amdgcn_s_barrier();  // k-block 0 is visible to every wave
    AMP_DMA_STAMP(1);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    // `younger`: k-blocks q + 2 .. that are in flight when k-block q + 1 is awaited (NS - 2 in the steady loop)
    auto kb = [&](const int q, const bool ahead, const int younger) {  // ahead: k-block q + NS - 1 exists and is issued here
      const unsigned char* sb = lds + (q % NS) * kStage;
      // R: fragments of the whole k-block; stage (q + NS - 1) % NS held k-block q - 1, whose last reads were retired in front of
      // a barrier this wave has passed
      read_w16(sb);
#pragma unroll
      for (int a = 0; a < 2 * TM; ++a) {
        y0[a] = *reinterpret_cast<const h8*>(sb + arow16 + a * 16 * 128 + ch16[0]);
        y1[a] = *reinterpret_cast<const h8*>(sb + arow16 + a * 16 * 128 + ch16[1]);
      }
      if (ahead) fill(q + NS - 1, (q + NS - 1) % NS);
      if (grp == 1) wait_younger(younger);  // group 1's pieces of k-block q + 1: before the barrier that ends this interval
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // M: 3 x 2 TM x 2 TN MFMAs (16 cycles each) between two barriers
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
        for (int b = 0; b < 2 * TN; ++b) c16[a][b] = mfma16(wf0[b], y1[a], c16[a][b]);
#pragma unroll
      for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
        for (int b = 0; b < 2 * TN; ++b) c16[a][b] = mfma16(wf1[b], y0[a], c16[a][b]);
#pragma unroll
      for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
        for (int b = 0; b < 2 * TN; ++b) c16[a][b] = mfma16(wf0[b], y0[a], c16[a][b]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (grp == 0) wait_younger(younger);  // group 0's pieces of k-block q + 1: behind its MFMAs
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    int q = 0;
    for (; q + NS - 1 < nq; ++q) kb(q, true, NS - 2);
    for (; q < nq; ++q) kb(q, false, nq - 2 - q > 0 ? nq - 2 - q : 0);  // the tail: nothing left to issue, fewer k-blocks in flight
  } else {
  if (first_tile) {
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
  } else {
    // a later tile of a persistent workgroup: its k-blocks 0 and 1 were issued in front of the previous tile's epilogue; its
    // stores and these fills retire out of order with each other, so the wait is for all of them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();  // k-block 0 is visible to every wave
  AMP_DMA_STAMP(1);
  if (grp == 1) __builtin_amdgcn_s_barrier();
  // one k-block = R0 M0 [R1 M1]; `second` = false drops the second k-step of the LAST k-block when it is zero padding only
  // (GemmF16Args::ksteps; uniform over the grid, so both wave groups drop the same two barriers; nothing is in flight then:
  // the last fill was drained a k-block earlier).  The steady loop stays free of extra control flow -- a `break` inside it
  // cost the 256 x 256 instantiations 350-420 B of scratch spills and 7x the time.
  auto kblock = [&](const int q, const bool second) {
    const unsigned char* sb = lds + (q & 1) * kStage;
    // R0: fragments of k-step 0; the other stage (k-block q - 1: its last reads were retired in front of a barrier
    // this wave has passed) takes k-block q + 1 (k-block 1 was issued in the prologue)
    if constexpr (MODE != 2) { read_w16(sb); read_x16(sb, 0); }
    else read_frags(sb, 0);
    if (q >= 1 && q + 1 < nq) fill(q + 1, (q + 1) & 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    matrix_segment(false, 0);
    if (!second) return;
    // R1: fragments of k-step 1 (MODE 1: of the lower TM 16-row blocks); group 1's pieces of k-block q + 1 must have landed
    // before the next barrier
    if constexpr (MODE != 2) read_x16(sb, 1);
    else read_frags(sb, 1);
    if (grp == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    matrix_segment(grp == 0, 1);  // group 0's pieces: behind its MFMAs
  };
  for (int q = 0; q + 1 < nq; ++q) kblock(q, true);
  kblock(nq - 1, MODE != 2 || 2 * nq - 1 < ksteps);  // (the 16 x 16 x 32 layers consume whole k-blocks: R1 / M1 are row halves there)
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  __syncthreads();  // every wave is done with the stages: the scratch below reuses them
  AMP_DMA_STAMP(2);

  if (MODE == 2) {
    // ---- plain fp32 product (training step).  Same register map as MODE 0: register r of lane (li, lh) is output ROW
    // (r & 3) + 8 (r >> 2) + 4 lh of the 32 x 32 block and COLUMN li, so a wave store covers two 128-B row segments.
    const float descale = 1.0f / (plane_scale(g.amax_a[0]) * plane_scale(g.amax_w[0]));  // powers of two: exact
    float* const C = g.C + (int64_t)slice * g.slice_stride;
    const int col0 = n0 + wn * (32 * TN);
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = col0 + b * 32 + li;
      const float bias = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * (32 * TM) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < g.M) {
            float v = acc[a][b][r] * descale;
            if (g.bias) v += bias;
            if (g.relu) v = fmaxf(v, 0.0f);
            if (g.mask) v = g.mask[row * g.ldmask + col] > 0.0f ? v : 0.0f;
            float* dst = C + row * g.ldc + col;
            if (g.accumulate) v += *dst;
            *dst = v;
          }
        }
    }
    return;
  }
  const LayerScales sc = layer_scales(g.range, g.amax, g.layer);
  const float descale = sc.descale;
  if (MODE == 0) {
    // Persistent workgroup (the grid is smaller than the tile count: launch_dma): move on to tile vb + gridDim.x.  Every
    // wave is past its last read of both stages (the __syncthreads above) and this epilogue does not touch the LDS, so
    // k-blocks 0 and 1 of the NEXT tile are issued here and land while this tile's hidden-layer rows are converted and
    // stored.  Measured per 256 x 256 tile of a 32 768-row launch (two tiles per CU, per-workgroup s_memrealtime stamps,
    // profiles/r03_l1_timeline.txt): start -> first k-step 2.5-5 us, k-loop 13.6-14.6 us, epilogue 4.6 us + 0.7-1.3 us of
    // store drain -- and 3.0 us between a workgroup's end and the start of its successor on the CU.
    const int64_t tile_m0 = m0;
    const int tile_n0 = n0;
    vb += gridDim.x;
    const bool more = f16_tile_of_block(g, vb, mt, nt);
    if (more) {
      m0 = tile_origin(mt);
      n0 = nt * BN;
      plan_fills();
      fill(0, 0);
      if (nq > 1) fill(1, 1);
    }
    // ---- layer-1 epilogue.  Accumulator register r of lane (i16, kq) is output ROW 4 kq + r of the 16-row block and, in
    // the MFMA of parity beta of a 32-column block, COLUMN rho(beta, i16) (see read_w16).  Per value relu_split4's arithmetic
    // (fma, max, two roundings to fp16: v_cvt_f16_f32 for p0 into the word's low half, ONE v_fma_mixhi_f16 for rn16(v - p0) --
    // exact difference, single rounding -- into its high half), then adjacent lanes swap words (one DPP move + one v_perm_b32):
    // even lanes hold the p0 halves of columns (rho(i), rho(i + 1)), odd lanes the p1 halves of (rho(i - 1), rho(i)) -- for
    // beta = 0 AND beta = 1, i.e. FOUR consecutive columns 2 i .. 2 i + 3 (i even) -- and ONE buffer_store_dwordx2 per
    // (row, 32-column k-block) writes the block's [p0 x 32 | p1 x 32] line: 128 contiguous bytes per 16 lanes, four rows per
    // instruction, no transpose through LDS.  The buffer is the wave's band of rows: base and extent in four SGPRs, a scalar
    // offset per (register, block), one per-lane offset for everything -- rows past M fall outside the extent and are dropped.
    const float s_h = sc.s_out, ds = descale * s_h;
    const uint32_t row_pitch = (uint32_t)(2 * g.ldh * (int64_t)sizeof(_Float16));
    const int band = __builtin_amdgcn_readfirstlane(wm) * (32 * TM);       // the wave's first row inside the tile
    const int col0 = tile_n0 + __builtin_amdgcn_readfirstlane(wn) * (32 * TN);  // ... and first column
    const int64_t rows_left = g.M - (tile_m0 + band);
    const int valid = rows_left <= 0 ? 0 : (rows_left < 32 * TM ? (int)rows_left : 32 * TM);
    unsigned char* const hband = reinterpret_cast<unsigned char*>(g.H) + (tile_m0 + band) * (int64_t)row_pitch + (col0 >> 5) * 128;
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(hband, 0, (int)((uint32_t)valid * row_pitch), 0x00020000);
    const uint32_t lane_off = (uint32_t)(4 * kq) * row_pitch + (uint32_t)((i16 & 1) * 64 + (i16 >> 1) * 8);
    const uint32_t sel = (i16 & 1) ? 0x03020706u : 0x05040100u;  // v_perm_b32(neighbour, own, sel): odd [nb.hi, own.hi], even [own.lo, nb.lo]
    float bs[TN][2];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int be = 0; be < 2; ++be) bs[b][be] = g.bias[col0 + b * 32 + 4 * (i16 >> 1) + (i16 & 1) + 2 * be] * s_h;
    typedef uint32_t uw2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          uw2 words;
#pragma unroll
          for (int be = 0; be < 2; ++be) {
            const float v = fmaxf(__builtin_fmaf(c16[a][2 * b + be][r], ds, bs[b][be]), 0.0f);
            uint32_t own;
            asm("v_cvt_f16_f32 %0, %1" : "=v"(own) : "v"(v));
            asm("v_fma_mixhi_f16 %0, -%0, 1.0, %1 op_sel_hi:[1,0,0]" : "+v"(own) : "v"(v));
            const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
            words[be] = __builtin_amdgcn_perm(nb, own, sel);
          }
          const int soff = (int)((uint32_t)(a * 16 + r) * row_pitch) + b * 128;  // wave-uniform
#ifdef AMP_L1_NO_STORES  // microbenchmark ablation (CONC0=1): everything but the stores; the words are kept alive by an empty asm
          asm volatile("" ::"v"(words[0]), "v"(words[1]));
#else
          __builtin_amdgcn_raw_buffer_store_b64(words, hrsrc, (int)lane_off, soff, AMP_L1_STORE_AUX);
#endif
        }
#ifdef AMP_DMA_TIMELINE
    AMP_DMA_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AMP_DMA_STAMP(4);
    if (g_dma_timeline && threadIdx.x == 0) g_dma_timeline[(size_t)stamp_row * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
#endif
    if (!more) return;
    first_tile = false;
#ifdef AMP_DMA_TIMELINE
    stamp_row = vb;
    AMP_DMA_STAMP(0);
#endif
    continue;
  }
  // ---- layer-2 epilogue: lane (i16, kq) holds activation row i16 of each 16-row block and output columns 4 kq + r of each
  //      16-column block; canonical partial logits, one per (row, 32-column block) = the sum of its two 16-column blocks'
  //      l2_partial16 (disc_gemm_f16.hpp) -- the same value whatever kernel / tile computed the accumulators
  const fv4* bias4 = reinterpret_cast<const fv4*>(g.bias + n0 + wn * (32 * TN) + 4 * kq);
  const fv4* w34 = reinterpret_cast<const fv4*>(g.w3 + n0 + wn * (32 * TN) + 4 * kq);
  float* red = reinterpret_cast<float*>(lds);  // [4 TN][BM]
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const fv4 bs0 = bias4[(2 * b) * 4], ws0 = w34[(2 * b) * 4], bs1 = bias4[(2 * b + 1) * 4], ws1 = w34[(2 * b + 1) * 4];
#pragma unroll
    for (int a = 0; a < 2 * TM; ++a) {
      const float v = l2_partial16(c16[a][2 * b], descale, bs0, ws0) + l2_partial16(c16[a][2 * b + 1], descale, bs1, ws1);
      if (kq == 0) red[(wn * TN + b) * BM + wm * (32 * TM) + a * 16 + i16] = v;
    }
  }
  __syncthreads();
  AMP_DMA_STAMP(3);
  constexpr int BPT = 4 * TN;  // 32-column blocks per tile
  const int n_blocks = g.N >> 5;
  for (int e = tid; e < BM * BPT; e += kDmaThreads) {
    const int r = e / BPT, j = e - r * BPT;
    const int64_t row = m0 + r;
    if (row < g.M) g.partial[row * n_blocks + nt * BPT + j] = red[j * BM + r];
  }
  return;
  }  // tile loop
}

}  // namespace amp
