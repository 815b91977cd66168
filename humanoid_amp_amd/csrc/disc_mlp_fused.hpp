// Layers 1 and 2 of the fp16-split discriminator forward as ONE kernel: the hidden layer never leaves the CU.
//
// Why.  Run as two GEMM launches, a 32 768-row chunk writes its hidden layer (4 B per element: 134 MB) and reads it straight
// back: 542 of the step's 884 MB of fabric traffic at 65 536 envs (profiles/pmc_traffic.json), and layer 1 -- k-loops + 134 MB
// of stores that do not overlap on this memory system (profiles/r03_l1_timeline.txt) -- is a third of the step at 0.30 of the
// fp16 peak.  Here a workgroup owns 128 rows x ALL 512 output columns; per 32 hidden units (one k-block of layer 2) every wave
//   A   computes the units' pre-activations for ITS 16 rows (layer 1: 2 x KX x 3 MFMAs, activation fragments resident in
//       registers for the whole tile, weight fragments from an LDS stage),
//       applies bias + ReLU + the fp16 plane split in registers -- the result IS the B operand of layer 2's MFMAs --
//   B   and accumulates it into its 16 x 512 slice of layer 2 (32 column blocks x 3 MFMAs, weight fragments from LDS).
// No hidden-layer stores, no hidden-layer fills, one launch instead of two (and the step's two 32 768-row chunks become one
// launch: chunking existed only to keep the hidden layer in the Infinity Cache).  Operand intake per workgroup and k-block:
// 64 KB of W2 + 24 KB of W1, all L2-resident (2.8 MB of planes) -- 21 B per matrix-pipe cycle of the CU, what the 256 x 256
// tiles take.
//
// Bit-identical to the two-kernel plans (tests/test_gpu_shard_equivalence.py, test_gemm_engines_vs_fp64 over every plan):
//   * layer 1: per accumulator the products (w0, x1), (w1, x0), (w0, x0) per k-block, k-blocks ascending -- the order of every
//     f16 kernel; the MFMA's own reduction depends on the k-slot alone, not on which row / lane of the tile an element sits in;
//   * the epilogue is relu_split4's arithmetic (fma, max, rn16, rn16 of the exact residual);
//   * the MFMA of layer 1 takes the WEIGHT rows as its A operand in the order unit(m, u) = 8 (m >> 2) + 4 u + (m & 3) (u = 0 / 1:
//     the two MFMAs of a 32-unit block; a per-lane row offset in the fill, nothing else), so that lane (i, kq) ends up with
//     units 8 kq .. 8 kq + 7 of activation row i in NATURAL order: exactly the B fragment of v_mfma_f32_16x16x32_f16 for layer 2;
//   * layer 2: (w0, h1), (w1, h0), (w0, h0) per k-block and accumulator, then the canonical partial logits (l2_partial16).
//
// Geometry: 512 threads = 8 waves.  Layer 1: wave w owns rows 16 w .. 16 w + 15 of the tile (48 registers of activation
// fragments: KX = 6 k-blocks x 2 planes) and hands its 16 x 32 slice of the hidden block to the workgroup through a 16-KB LDS
// tile in block layout (two ds_write_b128 per lane).  Layer 2: waves as 2 (rows) x 4 (columns), wave (wm, wn) owns rows
// 64 wm .. + 63 and columns 128 wn .. + 127: c16[4][8] = 128 accumulator registers, every weight fragment feeds 4 row blocks and
// every hidden fragment 8 column blocks (the first version gave every wave 16 rows x all 512 columns with the hidden block in
// registers: no LDS hand-over, but 88 KB of fragment reads per wave and k-block -- 170 B/clk of the LDS's 256 while the pipes
// run -- and the pipes were 55 % busy: 231 us per 65 536 rows; this layout reads 48 KB).  One workgroup per CU (140 KB of LDS).
// LDS: W1 slot KX x 4 KB | W2 half-slots (256 columns x 128 B = 32 KB) x 3 | hidden block 128 rows x 128 B | bias of layer 1 4 KB.
// A W2 half-slot holds, for every column group wn, 64 of its 128 columns (half 0: columns 128 wn .. + 63, half 1: the rest), so
// that every wave computes in every phase.
// A k-block q runs as three phases separated by one barrier each: A(q) (layer 1 -> hidden block), B1(q) on half-slot (2 q) % 3,
// B2(q) on half-slot (2 q + 1) % 3.  Fill schedule (every wave issues its share: 3 pieces of W1, 4 of a W2 half):
//   behind the barrier that ends A(q)  : W1(q + 1)  into the W1 slot (A(q) was its last reader)
//   behind the barrier that ends B1(q) : W2b(q + 1) into the half-slot B1(q) just left
//   behind the barrier that ends B2(q) : W2a(q + 2) into the half-slot B2(q) just left
// and in front of the barrier that ends A(q) / B1(q) / B2(q) a wave waits (counted vmcnt: 11 / 7 / 4 younger pieces stay in
// flight) for ITS pieces of W2a(q) / W2b(q) / W1(q + 1) -- the data of the NEXT phase: the LDS-DMA visibility rule
// (MI355X_MICROARCH.md: the issuing wave's vmcnt + a barrier the reader has passed); every refill is issued behind a barrier
// that follows the lgkmcnt(0) of the slot's last reads, and the hidden block is rewritten by A(q + 1) behind the barrier that
// ends B2(q) (its fragments are read once, at the head of B1(q), and stay in registers).
#pragma once
#include "disc_gemm_f16_dma.hpp"

namespace amp {

struct FusedArgs {
  const _Float16* X; int64_t ldx; int64_t M;   // scaled input, block layout: row pitch 2 * ldx halves (ldx = 32 KX)
  const _Float16* W1b;                         // [h1][KX][2][32] planes of s_w1 W1
  const _Float16* W2b;                         // [512][h1 / 32][2][32] planes of s_w2 W2
  const float* b1; const float* b2; const float* w3;
  const DiscRange* range; const float* amax;
  int32_t h1;                                  // hidden units (multiple of 32, <= 1024)
  float* partial;                              // [M, 16] canonical partial logits
  // RAWX instantiation: the activation fragments come straight from the fp32 observation rows -- the kernel applies the scaler,
  // the clamp and the plane split to its 8 KX elements per lane itself (the arithmetic of disc_scale_split_kernel / of the env
  // step's fused scaler, operation for operation: the same bits), so no scaled copy of the input is ever written
  const float* raw; int64_t raw_ld;            // [M, raw_cols] fp32 rows, pitch raw_ld floats (even; rows 8-B aligned)
  int32_t raw_cols;                            // K D (even); columns in [raw_cols, 32 KX) are zero
  const float* mean; const float* den;         // scaler vectors (fp32, >= raw_cols entries)
  float clip, s_x;                             // the clamp and the input planes' scale
};

constexpr int kFusedThreads = 512, kFusedRows = 128, kFusedN2 = 512;
template <int KX>
struct FusedLds {
  static constexpr int kW1Slot = KX * 4096, kW2Half = 256 * 128;
  // (the small regions first: their addresses are 16-bit immediate offsets off the lane's two chunk registers)
  static constexpr int kW1 = 0, kH = kW1Slot, kBias = kH + kFusedRows * 128, kW2 = kBias + 4096, kBytes = kW2 + 3 * kW2Half;
};

#ifdef AMP_FUSED_TIMELINE  // diagnostic builds only (tools/fused_timeline.py): per-workgroup stamps around the k-loop
__device__ unsigned long long* g_fused_timeline;  // [workgroups][4]: s_memtime / s_memrealtime at loop start, at loop end
extern "C" int amp_debug_fused_timeline(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fused_timeline), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#define AMP_FUSED_STAMP(slot)                                                                    \
  do {                                                                                           \
    if (g_fused_timeline && threadIdx.x == 0) {                                                  \
      g_fused_timeline[(size_t)blockIdx.x * 4 + 2 * (slot)] = __builtin_amdgcn_s_memtime();      \
      g_fused_timeline[(size_t)blockIdx.x * 4 + 2 * (slot) + 1] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                            \
  } while (0)
#else
#define AMP_FUSED_STAMP(slot) do {} while (0)
#endif
template <int KX, bool RAWX = false>
__global__ __launch_bounds__(kFusedThreads, 2) void disc_mlp_fused_kernel(FusedArgs g) {
  using L = FusedLds<KX>;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int i16 = lane & 15, kq = lane >> 4, swz16 = (i16 >> 1) & 7;
  const int nq = g.h1 >> 5;  // k-blocks of layer 2 = 32-unit blocks of layer 1
  // the last row tile of a ragged M is moved up to end at M (its overlap with the tile above is computed twice from the same
  // operands: the same bits land twice); M >= 128 is the launcher's precondition
  int64_t m0 = (int64_t)blockIdx.x * kFusedRows;
  if (m0 + kFusedRows > g.M) m0 = g.M - kFusedRows;

  // ---- activation fragments of the wave's 16 layer-1 rows: B operand, lane (i, kq) = row i, k = 32 kbx + 8 kq .. + 7 ----------
  h8 x[KX][2];
  // RAWX: the fp32 row pieces, scaler means and denominators of the lane's 8 KX columns are REQUESTED here (column pairs (c, c + 1),
  // c even: one 8-B load each; K D is even, so a pair is wholly inside or outside the row) and converted behind the first stage
  // fills' issue, so that the ~1 000 VALU operations of the conversion run under the fills' latency
  typedef float f2_t __attribute__((ext_vector_type(2)));
  f2_t rv[RAWX ? KX : 1][4], rmu[RAWX ? KX : 1][4], rdn[RAWX ? KX : 1][4];
  if constexpr (RAWX) {
    const float* rr = g.raw + (m0 + 16 * wave + i16) * g.raw_ld + 8 * kq;
#pragma unroll
    for (int kb = 0; kb < KX; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 32 * kb + 8 * kq + 2 * j;
        const bool in = c < g.raw_cols;
        rv[kb][j] = in ? *reinterpret_cast<const f2_t*>(rr + 32 * kb + 2 * j) : f2_t{0.0f, 0.0f};
        rmu[kb][j] = in ? *reinterpret_cast<const f2_t*>(g.mean + c) : f2_t{0.0f, 0.0f};
        rdn[kb][j] = in ? *reinterpret_cast<const f2_t*>(g.den + c) : f2_t{1.0f, 1.0f};
      }
  } else {
    const _Float16* xr = g.X + (m0 + 16 * wave + i16) * (2 * g.ldx) + 8 * kq;
#pragma unroll
    for (int kb = 0; kb < KX; ++kb) {
      x[kb][0] = *reinterpret_cast<const h8*>(xr + kb * 64);
      x[kb][1] = *reinterpret_cast<const h8*>(xr + kb * 64 + 32);
    }
  }
  // ---- fill plan ------------------------------------------------------------------------------------------------------------
  // W1(q): LDS sub-stage kbx = 32 rows x 128 B, LDS row rho = 16 u + i holds unit 32 q + 8 (i >> 2) + 4 u + (i & 3); 4 pieces of
  // 8 rows per kbx, 4 KX pieces per k-block: wave w issues pieces 3 w .. 3 w + 2 (KX = 6).  W2 half hf (256 LDS rows): LDS row
  // rho = 64 wn' + r holds column 128 wn' + 64 hf + r; 32 pieces, wave w issues 4 w .. 4 w + 3.  Lane l of a piece: row
  // 8 j + (l >> 3), stored chunk l & 7 = source chunk (l & 7) ^ ((row >> 1) & 7).
  // KX = 5 (20 pieces on 8 waves x 3): the last four slots re-issue pieces 16 .. 19 -- the same bytes to the same LDS addresses a
  // second time -- so that every wave issues the same NUMBER of pieces and the counted vmcnt waits stay wave-independent.
  constexpr int NP1 = (4 * KX + 7) / 8, NP2 = 4;
  uint32_t off1[NP1], off2[2];  // off2: pieces t and t + 2 differ by 16 rows (same swizzle): a wave-uniform + 16 pitch2 on the base
  int dst1[NP1];
  const uint32_t pitch1 = KX * 128u, pitch2 = (uint32_t)nq * 128u;  // bytes per weight row (block layout)
#pragma unroll
  for (int t = 0; t < NP1; ++t) {
    int p = wave * NP1 + t;
    if (p >= 4 * KX) p -= 4;
    const int kbx = p >> 2, j = p & 3;
    const int rho = 8 * j + (lane >> 3), u = rho >> 4, i = rho & 15;
    const int unit = 8 * (i >> 2) + 4 * u + (i & 3);
    const int c = (lane & 7) ^ ((rho >> 1) & 7);
    off1[t] = (uint32_t)unit * pitch1 + (uint32_t)kbx * 128u + 16u * (uint32_t)c;
    dst1[t] = kbx * 4096 + j * 1024;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rho = 8 * (wave * NP2 + t) + (lane >> 3);  // rows 32 w .. 32 w + 31 of the half-slot: inside one 64-row column group
    const int col = 128 * (rho >> 6) + (rho & 63);  // + 64 hf
    const int c = (lane & 7) ^ ((rho >> 1) & 7);
    off2[t] = (uint32_t)col * pitch2 + 16u * (uint32_t)c;
  }
  const unsigned char* const w1g = reinterpret_cast<const unsigned char*>(g.W1b);
  const unsigned char* const w2g = reinterpret_cast<const unsigned char*>(g.W2b);
  auto fill_w1 = [&](const int q) {  // units 32 q .. 32 q + 31
    const unsigned char* src = w1g + (int64_t)q * (32 * (int64_t)pitch1);
    unsigned char* dst = lds + L::kW1;
#pragma unroll
    for (int t = 0; t < NP1; ++t) __builtin_amdgcn_global_load_lds((gptr_t)(src + off1[t]), (lptr_t)(dst + dst1[t]), 16, 0, 0);
  };
  auto fill_w2 = [&](const int q, const int half, const int hs) {  // k-block q of the half's columns -> half-slot hs
    const unsigned char* src = w2g + (int64_t)half * (64 * (int64_t)pitch2) + (int64_t)q * 128;
    unsigned char* dst = lds + L::kW2 + hs * L::kW2Half + wave * (NP2 * 1024);
#pragma unroll
    for (int t = 0; t < NP2; ++t)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + (t >> 1) * (16 * (int64_t)pitch2) + off2[t & 1]), (lptr_t)(dst + t * 1024), 16, 0, 0);
  };
  static_assert(KX >= 4 && KX <= 6 && 8 * NP1 - 4 * KX <= 4 && NP2 == 4, "piece plan: 8 waves x NP1 slots cover the 4 KX pieces (+ <= 4 repeats)");

  // prologue: W2a(0), W1(0) -- then the steady issue order W2b(q), W2a(q + 1), W1(q + 1) of the interval heads.  Everything the
  // first phase needs is REQUESTED before anything is awaited (stages, activation fragments above, the raw bias, the range
  // record): one memory round trip in front of the first MFMA instead of four dependent ones (range record -> bias -> LDS ->
  // stages: ~2 us per tile, measured with tools/fused_timeline.py: tiles of the second round used to start 97-103 us after the
  // first for a 91.5-us k-loop).
  fill_w2(0, 0, 0);
  fill_w1(0);
  float b1raw[2];  // h1 <= 1024: at most two units per lane
#pragma unroll
  for (int u = 0; u < 2; ++u) b1raw[u] = tid + u * kFusedThreads < g.h1 ? g.b1[tid + u * kFusedThreads] : 0.0f;
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (RAWX) {
    // scaler, clamp, plane split: the arithmetic of disc_scale_split_kernel and of the env step's fused scaler, operation for operation
    const float clip = g.clip, s_x = g.s_x;
#pragma unroll
    for (int kb = 0; kb < KX; ++kb)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float t = ((i & 1) ? rv[kb][i >> 1].y : rv[kb][i >> 1].x) - ((i & 1) ? rmu[kb][i >> 1].y : rmu[kb][i >> 1].x);
        t = t / ((i & 1) ? rdn[kb][i >> 1].y : rdn[kb][i >> 1].x);   // skrl RunningStandardScaler: exact fp32 divide
        t = fminf(fmaxf(t, -clip), clip);
        const float y = t * s_x;
        const _Float16 a = (_Float16)y;
        x[kb][0][i] = a;
        x[kb][1][i] = (_Float16)(y - (float)a);
      }
  }
  const LayerScales sc1 = layer_scales(g.range, g.amax, 1);
  const float s_h = sc1.s_out, ds = sc1.descale * s_h;
  const float descale2 = layer_scales(g.range, g.amax, 2).descale;
  // layer 1's bias, scaled, in LDS: lane (i, kq) needs the eight units 32 q + 8 kq .. + 7 per k-block
  float* const b1s = reinterpret_cast<float*>(lds + L::kBias);
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (tid + u * kFusedThreads < g.h1) b1s[tid + u * kFusedThreads] = b1raw[u] * s_h;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // W1(0), W2a(0) and the bias are visible to every wave

  fx4 c16[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) c16[a][b] = fx4{0.0f, 0.0f, 0.0f, 0.0f};
  const int ch0 = (kq ^ swz16) * 16, ch1 = ((4 + kq) ^ swz16) * 16;  // plane 0 / 1 chunk of the lane inside a 128-B row
  const int frag_row = i16 * 128;
  unsigned char* const hrow = lds + L::kH + (16 * wave) * 128 + frag_row;           // the lane's row of the hidden block (writer)
  const unsigned char* const hfrag = lds + L::kH + (64 * wm) * 128 + frag_row;      // ... and of the wave's layer-2 rows (reader)
  const int w2frag = (64 * wn) * 128 + frag_row;
  h8 hf0[4], hf1[4];  // hidden fragments of the wave's four 16-row blocks (read at the head of B1, used by B1 and B2)

  // A(q): layer 1 for units 32 q .. 32 q + 31 of the wave's 16 rows -> the wave's rows of the hidden block
  auto phase_a = [&](const int q) {
    const unsigned char* s1 = lds + L::kW1 + frag_row;
    const fv4 bia0 = *reinterpret_cast<const fv4*>(b1s + 32 * q + 8 * kq), bia1 = *reinterpret_cast<const fv4*>(b1s + 32 * q + 8 * kq + 4);
    fx4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = {0.0f, 0.0f, 0.0f, 0.0f};
#ifndef AMP_FUSED_NO_PRIO_A
    // Phase A is a serial path: 36 MFMAs, THEN ~60 VALU operations and two LDS writes that depend on them.  The SIMD's other
    // wave is in a layer-2 phase (48 independent MFMAs): with equal priority the pipe alternates between the two, this wave's
    // MFMAs finish when the partner has a dozen left, and the epilogue runs beside an idle pipe.  Raised priority for the MFMA part
    // puts this wave's 36 first; its epilogue then runs under the partner's remaining MFMAs.
    __builtin_amdgcn_s_setprio(2);
#endif
#if defined(AMP_FUSED_XP_W1_READS)
    // ABLATION builds only (tools/build_variant.sh ... -DAMP_FUSED_XP_W1_READS=2 / 6; results are WRONG): the W1 fragments are read
    // for every AMP_FUSED_XP_W1_READS-th k-block only and reused for the others -- the same MFMAs on half (or a sixth of) layer 1's
    // LDS fragment reads: what a kernel whose W1 fragments fed two (six) row blocks could gain at most.
    h8 w00{}, w01{}, w10{}, w11{};
#endif
#pragma unroll
    for (int kb = 0; kb < KX; ++kb) {
#if defined(AMP_FUSED_XP_W1_READS)
      if (kb % AMP_FUSED_XP_W1_READS == 0) {
        w00 = *reinterpret_cast<const h8*>(s1 + kb * 4096 + ch0); w01 = *reinterpret_cast<const h8*>(s1 + kb * 4096 + ch1);
        w10 = *reinterpret_cast<const h8*>(s1 + kb * 4096 + 2048 + ch0); w11 = *reinterpret_cast<const h8*>(s1 + kb * 4096 + 2048 + ch1);
      }
#else
      const h8 w00 = *reinterpret_cast<const h8*>(s1 + kb * 4096 + ch0), w01 = *reinterpret_cast<const h8*>(s1 + kb * 4096 + ch1);
      const h8 w10 = *reinterpret_cast<const h8*>(s1 + kb * 4096 + 2048 + ch0), w11 = *reinterpret_cast<const h8*>(s1 + kb * 4096 + 2048 + ch1);
#endif
      a0 = mfma16(w00, x[kb][1], a0);
      a1 = mfma16(w10, x[kb][1], a1);
      a0 = mfma16(w01, x[kb][0], a0);
      a1 = mfma16(w11, x[kb][0], a1);
      a0 = mfma16(w00, x[kb][0], a0);
      a1 = mfma16(w10, x[kb][0], a1);
    }
#ifndef AMP_FUSED_NO_PRIO_A
    __builtin_amdgcn_s_setprio(0);
#endif
    // (the same arithmetic in ~40 instead of ~70 instructions -- v_cvt_f16_f32 + v_fma_mixhi_f16 + v_perm packing -- was measured
    //  equal once the phase's MFMAs run at raised priority: 203.4 us either way)
    h4 p0a, p1a, p0b, p1b;
    relu_split4(a0, ds, bia0, p0a, p1a);
    relu_split4(a1, ds, bia1, p0b, p1b);
    *reinterpret_cast<h8*>(hrow + ch0) = h8{p0a[0], p0a[1], p0a[2], p0a[3], p0b[0], p0b[1], p0b[2], p0b[3]};
    *reinterpret_cast<h8*>(hrow + ch1) = h8{p1a[0], p1a[1], p1a[2], p1a[3], p1b[0], p1b[1], p1b[2], p1b[3]};
  };
  auto read_h = [&]() {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      hf0[a] = *reinterpret_cast<const h8*>(hfrag + a * 2048 + ch0);
      hf1[a] = *reinterpret_cast<const h8*>(hfrag + a * 2048 + ch1);
    }
  };
  // B1 / B2: layer 2, one k-block, the first / second 64 columns of the wave's 128 from half-slot hs
  // `refill()` issues the wave's share of the interval's refill: behind the FIRST column block's MFMAs, not at the head of the
  // interval -- in I2 both waves of a SIMD are in a layer-2 phase, and with the pieces in front (address arithmetic + ~100
  // cycles of issue each) neither reached its first MFMA for ~600 cycles behind the barrier.
  auto phase_b = [&](auto half_c, const int hs, auto&& refill) {
    constexpr int half = decltype(half_c)::value;
    const unsigned char* s2 = lds + L::kW2 + hs * L::kW2Half + w2frag;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const h8 w0 = *reinterpret_cast<const h8*>(s2 + cb * 2048 + ch0), w1 = *reinterpret_cast<const h8*>(s2 + cb * 2048 + ch1);
#pragma unroll
      for (int a = 0; a < 4; ++a) c16[a][4 * half + cb] = mfma16(w0, hf1[a], c16[a][4 * half + cb]);
#pragma unroll
      for (int a = 0; a < 4; ++a) c16[a][4 * half + cb] = mfma16(w1, hf0[a], c16[a][4 * half + cb]);
#pragma unroll
      for (int a = 0; a < 4; ++a) c16[a][4 * half + cb] = mfma16(w0, hf0[a], c16[a][4 * half + cb]);
      if (cb == 0) {
        __builtin_amdgcn_sched_barrier(0);
        refill();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto no_refill = []() {};
  using half0_t = std::integral_constant<int, 0>;
  using half1_t = std::integral_constant<int, 1>;
  auto next3 = [](const int v) { return v == 2 ? 0 : v + 1; };

  // Two wave groups, one wave of each per SIMD, run the phase sequence A B1 B2 one interval apart (group 1 = waves 4-7 = the
  // tile's rows 64-127 in BOTH layers: a group's hidden rows are written and read by the group itself), so that a SIMD always
  // pairs two DIFFERENT phases -- layer 1's dependent epilogue (VALU, LDS write) runs under the partner's layer-2 MFMAs instead
  // of under its own copy of the same epilogue.  Interval I0(q): G0 A(q) | G1 B2(q - 1);  I1(q): G0 B1(q) | G1 A(q);
  // I2(q): G0 B2(q) | G1 B1(q); one workgroup barrier per interval.  A stage is read in two consecutive intervals (G0, then G1)
  // and refilled at the head of the interval after them, by every wave:
  //   head of I0(q): W2b(q)     -> half-slot (2 q + 1) % 3   (last readers: W2a(q - 1) in I1 / I2 of q - 1)
  //   head of I1(q): W2a(q + 1) -> half-slot (2 q + 2) % 3   (W2b(q - 1): I2(q - 1), I0(q))
  //   head of I2(q): W1(q + 1)  -> the W1 slot               (W1(q): I0(q), I1(q))
  // and awaited in front of the barrier that ends the interval BEFORE the first use: end of I0(q): W2a(q) (7 younger pieces stay
  // in flight), end of I1(q): W2b(q) (4), end of I2(q): W1(q + 1) (0).
  // The loop body is the same for every k-block (no first / last special cases: a peeled or unswitched copy of 264 MFMAs costs
  // registers and instruction cache): group 1's B2(-1) of the first interval runs on ZERO hidden fragments over the landed W2a(0)
  // (adds exact zeros), and the last k-block's look-ahead fills re-fetch k-block nq - 1 into slots nobody reads again.
#pragma unroll
  for (int a = 0; a < 4; ++a) hf0[a] = hf1[a] = h8{0, 0, 0, 0, 0, 0, 0, 0};
  // (Two loops, one per group -- the group is wave-uniform, and in ONE loop with a branch per interval the compiler has to keep
  //  hf0 / hf1 alive through the other group's phase A: 420 B of scratch.  Both loops execute the same barriers.)
  // younger pieces that stay in flight behind the awaited stage: W1(q) + W2b(q) behind W2a(q) (NP1 + NP2 = 7 at KX = 5 / 6, 6 at
  // KX = 4), W2a(q + 1) behind W2b(q) (NP2 = 4), nothing behind W1(q + 1)
  auto end_interval = [&](auto younger_c) {
    constexpr int Y = decltype(younger_c)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(Y) : "memory");
    __builtin_amdgcn_s_barrier();
  };
  using y7 = std::integral_constant<int, NP1 + NP2>;
  using y4 = std::integral_constant<int, NP2>;
  using y0 = std::integral_constant<int, 0>;
  int m3 = 0;  // (2 q) % 3
  AMP_FUSED_STAMP(0);
  if (wm == 0) {
#pragma clang loop unroll(disable)
    for (int q = 0; q < nq; ++q) {
      const int qn = q + 1 < nq ? q + 1 : q;              // the look-ahead k-block (clamped at the end)
      const int sA = m3, sB = next3(m3), sC = next3(sB);  // half-slots of W2a(q), W2b(q), and of W2b(q - 1) = W2a(q + 1)
#ifdef AMP_FUSED_FILL_FIRST
      fill_w2(q, 1, sB);   // I0(q)
      phase_a(q);
#else
      phase_a(q);          // I0(q): the serial phase starts right behind the barrier; its share of the refill goes out at its end
      fill_w2(q, 1, sB);
#endif
      end_interval(y7{});
      read_h();            // I1(q)
      phase_b(half0_t{}, sA, [&]() { fill_w2(qn, 0, sC); });
      end_interval(y4{});
      phase_b(half1_t{}, sB, [&]() { fill_w1(qn); });   // I2(q)
      end_interval(y0{});
      m3 = sC;
    }
  } else {
#pragma clang loop unroll(disable)
    for (int q = 0; q < nq; ++q) {
      const int qn = q + 1 < nq ? q + 1 : q;
      const int sA = m3, sB = next3(m3), sC = next3(sB);
      phase_b(half1_t{}, q > 0 ? sC : sA, [&]() { fill_w2(q, 1, sB); });   // I0(q)
      end_interval(y7{});
#ifdef AMP_FUSED_FILL_FIRST
      fill_w2(qn, 0, sC);  // I1(q)
      phase_a(q);
#else
      phase_a(q);          // I1(q)
      fill_w2(qn, 0, sC);
#endif
      end_interval(y4{});
      read_h();            // I2(q)
      phase_b(half0_t{}, sA, [&]() { fill_w1(qn); });
      end_interval(y0{});
      m3 = sC;
    }
    phase_b(half1_t{}, next3(next3(m3)), no_refill);  // B2(nq - 1): half-slot (2 nq - 1) % 3
  }

  AMP_FUSED_STAMP(1);
  // ---- canonical partial logits: one per (row, 32-column block) = l2_partial16 of its two 16-column blocks.  The wave's 128
  //      columns are the 32-column blocks 4 wn .. 4 wn + 3 (its column blocks 2 t, 2 t + 1); every lane of a row ends up with the
  //      same four values, the lanes of group kq = 0 store them (one 16-B store per row) -----------------------------------------
  fv4 bs[8], ws[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    bs[b] = *reinterpret_cast<const fv4*>(g.b2 + 128 * wn + 16 * b + 4 * kq);
    ws[b] = *reinterpret_cast<const fv4*>(g.w3 + 128 * wn + 16 * b + 4 * kq);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    fv4 out;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      out[t] = l2_partial16(c16[a][2 * t], descale2, bs[2 * t], ws[2 * t]) + l2_partial16(c16[a][2 * t + 1], descale2, bs[2 * t + 1], ws[2 * t + 1]);
    const int64_t row = m0 + 64 * wm + 16 * a + i16;
    if (kq == 0) *reinterpret_cast<fv4*>(g.partial + row * 16 + 4 * wn) = out;
  }
}

}  // namespace amp
