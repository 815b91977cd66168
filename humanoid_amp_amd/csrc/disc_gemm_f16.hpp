// Split-operand GEMM of the discriminator forward: fp32 accuracy out of the fp16 matrix pipe.
//
// Every fp32 operand x (activations and weights alike) is carried as two fp16 planes of s*x, s a power of two that
// brings the tensor's bound to <= 2^15:
//     p0 = rn16(s x),  p1 = rn16(s x - p0)        (s x - p0 is exact in fp32)   =>   |s x - p0 - p1| <= 2^-23 |s x|
// for every element within 2^18 of the bound (smaller ones keep an ABSOLUTE error <= 2^-25, i.e. 2^-40 of the bound:
// fp16 subnormals are honoured by the gfx950 MFMA).  A product then needs three v_mfma_f32_32x32x16_f16 into ONE fp32
// accumulator -- fp16 x fp16 products are exact in it --
//     acc += a0 b0 + a0 b1 + a1 b0            a.b = acc / (s_a s_b)
// and drops only a1 b1 (2^-22 relative): 96 MFMA cycles per 32x32x16 block instead of the 512 of eight
// v_mfma_f32_32x32x2_f32, at a measured error no larger than the fp32 fma chain's (tests/test_gpu_disc.py).
// In GEMM terms it is a plain fp16 product over the concatenated reduction [a0 | a0 | a1] . [b0 | b1 | b0] that
// stages only two planes per operand.
//
//   MODE 0: H = relu(A W^T / (s_a s_w) + bias), written as the two fp16 planes of s_h H (the next layer's operand)
//   MODE 1: partial[m, nt] = sum over the tile's columns of relu(A W^T / (s_a s_w) + bias)[m, n] * w3[n]
//
// Tile: (64 TM) x (64 TN) x BK, 4 waves as 2 x 2, each wave TM x TN accumulator blocks.  The MFMA's first operand is
// the WEIGHT fragment, so the accumulators hold the transposed tile (registers = output columns n, lanes =
// activation rows m): bias / w3 are per-register constants and the 512 -> 1 layer is a per-lane sum.
// One LDS stage (rows padded by 8 halves: conflict-free ds_read_b128), register prefetch of the next k-tile pinned
// above the MFMAs, raw lgkmcnt-only barriers -- the recipe the fp32 kernel (disc_gemm.hpp) arrived at.
#pragma once
#include "amp_common.hpp"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float fx16 __attribute__((ext_vector_type(16)));
typedef float fv4 __attribute__((ext_vector_type(4)));
typedef uint32_t uv4 __attribute__((ext_vector_type(4)));

namespace amp {

// Range record of the discriminator (device resident; disc.hip fills it whenever weights or scaler change).
struct DiscRange {
  float s_w1, s_w2;    // power-of-two plane scales of W1 / W2
  float wsum1, bmax1;  // max_j sum_k |W1[j, k]|, max_j |b1[j]|: |H1| <= wsum1 * bound_x + bmax1
  float clip;          // static bound of the scaled input (the scaler's clamp)
  float pad[3];
  // disc_weight_range_kernel's accumulators: max |W1|, max |W2|, max row sum, max |b1| as uint bit patterns (atomicMax) and the
  // ticket of its workgroups; all zero between launches (the workgroup with the last ticket publishes the fields above and
  // clears them: no memset in front of the launch)
  unsigned raw[4];
  unsigned ticket;
  unsigned pad2[3];
};
static_assert(sizeof(DiscRange) == 64, "DiscRange is one 64-B record");

// largest power of two s with s * bound < 2^15 (fp16 tops out at 65504)
__host__ __device__ __forceinline__ float plane_scale(float bound) {
  if (!(bound > 0.0f)) return 1.0f;
  int e;
  (void)frexpf(bound, &e);  // bound = m 2^e, m in [0.5, 1)
  return ldexpf(1.0f, 15 - e);
}

// Per-layer scales, recomputed by every workgroup from the range record (a handful of scalar ops; keeps the forward
// free of per-call state in the handle): layer 1 consumes planes of s_x Xs and s_w1 W1 and emits planes of s_h H1,
// layer 2 consumes those and s_w2 W2.  bound_x = the dynamic abs-max when `amax` is given, else the clamp.
struct LayerScales { float descale, s_out; };
__device__ __forceinline__ LayerScales layer_scales(const DiscRange* r, const float* amax, int layer) {
  const float bx = amax ? amax[0] : r->clip;
  const float s_h = plane_scale(r->wsum1 * bx + r->bmax1);
  LayerScales o;
  if (layer == 1) {
    o.descale = 1.0f / (plane_scale(bx) * r->s_w1);
    o.s_out = s_h;
  } else {
    o.descale = 1.0f / (s_h * r->s_w2);
    o.s_out = 0.0f;
  }
  return o;
}

struct GemmF16Args {
  // activations: planes A + p * plane_a with rows of lda halves, or (BLOCKS) block layout: rows of 2 * lda halves, per
  // k-block of 32 values [p0 x 32 | p1 x 32] -- the layout the scaled input is produced in (amp_env_step's fused scaler)
  const _Float16* A; int64_t lda, plane_a; int64_t M;
  const _Float16* W; int64_t plane_w; int32_t Kp; int32_t N;    // weight planes [N, Kp]
  const float* bias;
  const DiscRange* range; const float* amax; int32_t layer;     // -> layer_scales()
  _Float16* H; int64_t ldh, plane_h;                            // mode 0 output planes
  const float* w3; float* partial;                              // mode 1 output [M, n_tiles]
  int32_t n_tiles, m_tiles;
  // k-steps (of 16) that can hold non-zero values: ceil(true K / 16).  Kp pads K to the k-block (32) with zeros in BOTH
  // operands; a k-step made of padding only adds exact zeros to every accumulator and is skipped (K = 166 -> 11 of 12
  // k-steps: 8 % of layer 1's MFMAs).  0 = all of Kp / 16.
  int32_t ksteps;
  // MODE 2 of the LDS-DMA kernel (training step: C = A W^T as a plain fp32 product; disc_train.hip).  A and W are planes
  // of s_a A and s_w W with s = plane_scale(abs-max of the operand): the kernel reads the two abs-max words itself.
  float* C; int64_t ldc;                       // fp32 output
  const float* amax_a; const float* amax_w;    // device abs-max of A / W
  const float* mask; int64_t ldmask;           // optional gate: C = (mask > 0) ? C : 0
  int32_t relu;                                // C = max(C + bias, 0) (bias != nullptr: C + bias)
  int32_t accumulate;                          // C += (not with k_slices > 1)
  int32_t k_slices; int64_t slice_stride;      // split-K: slice s reduces its run of k-blocks into C + s * slice_stride
};

// rn16(v), rn16(v - rn16(v)) for four values
__device__ __forceinline__ void split_planes4(const fv4 v, h4& p0, h4& p1) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 a = (_Float16)v[i];
    p0[i] = a;
    p1[i] = (_Float16)(v[i] - (float)a);
  }
}

// Hidden-layer epilogue for four accumulator values: the planes of max(acc * descale + bias, 0) * s_h.  descale and s_h
// are powers of two, so this equals max(fma(acc, descale * s_h, bias * s_h), 0) bit for bit (scaling by a power of two
// commutes with rounding) -- `ds` = descale * s_h, `bs` = bias * s_h; and v - p0 is exact in fp32, so the low plane is one
// mixed-precision fma rounded once to fp16.  4 VALU operations per value instead of 10: at K = 192 the epilogue's
// conversions cost as many SIMD cycles as the layer's MFMAs.
__device__ __forceinline__ void relu_split4(const fv4 acc, const float ds, const fv4 bs, h4& p0, h4& p1) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = fmaxf(__builtin_fmaf(acc[i], ds, bs[i]), 0.0f);
    const _Float16 a = (_Float16)v;
    p0[i] = a;
    p1[i] = (_Float16)__builtin_fmaf((float)a, -1.0f, v);
  }
}

// planes of scale[0] * src[rows, cols] (row pitch ld_src floats) -> dst + p * plane, rows of ld_dst halves; columns in
// [cols, ld_dst) are zero.  One thread per four columns.
static __global__ __launch_bounds__(kBlock) void split_rows_f16_kernel(const float* __restrict__ src, int64_t rows, int cols,
                                                                int64_t ld_src, const float* __restrict__ scale,
                                                                _Float16* __restrict__ dst, int64_t ld_dst, int64_t plane) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x, per_row = ld_dst / 4;
  if (q >= rows * per_row) return;
  const int64_t r = q / per_row;
  const int c = (int)(q - r * per_row) * 4;
  const float s = scale[0];
  fv4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = c + i < cols ? src[r * ld_src + c + i] * s : 0.0f;
  h4 p0, p1;
  split_planes4(v, p0, p1);
  *reinterpret_cast<h4*>(&dst[r * ld_dst + c]) = p0;
  *reinterpret_cast<h4*>(&dst[plane + r * ld_dst + c]) = p1;
}

// `b`: the (virtual) block index -- blockIdx.x, or blockIdx.x + i * gridDim.x for the i-th tile of a persistent workgroup
// (gridDim.x is a multiple of 8, so every tile of a workgroup belongs to the XCD the workgroup runs on)
__device__ __forceinline__ bool f16_tile_of_block(const GemmF16Args& g, unsigned b, int& mt, int& nt) {
  const int total = g.m_tiles * g.n_tiles;
  const int per_xcd = (total + 7) / 8;
  const int in_xcd = (int)(b >> 3);
  const int v = (int)(b & 7) * per_xcd + in_xcd;  // one XCD (one L2) owns a run of row tiles
  if (in_xcd >= per_xcd || v >= total) return false;
  mt = v / g.n_tiles;
  nt = v - mt * g.n_tiles;
  return true;
}
__device__ __forceinline__ bool f16_tile_of_block(const GemmF16Args& g, int& mt, int& nt) {
  return f16_tile_of_block(g, blockIdx.x, mt, nt);
}

// ---- layer 2 on v_mfma_f32_16x16x32_f16 ------------------------------------------------------------------------------
// On MI355X the matrix pipes are power-limited on real operands, and two waves per SIMD sustain 13 % more FLOP/s with the
// 16 x 16 x 32 shape than with 32 x 32 x 16 (bare streams 1 820-1 836 vs 1 617-1 626 TFLOP/s at 1.85 vs 1.61 GHz,
// tools/mfma_shape_power.hip; inside the 256 x 256 layer-2 kernel 78.8 -> 69.3 us per 32 768 rows).  Every layer-2 (MODE 1)
// kernel of the fp16 engine therefore uses it -- all of them, so that a row's logit stays independent of the kernel its shard
// size selects (the instruction's own reduction over its 32 k-values is part of the result).  Operand / result layout of
//   D[i][j] (16 x 16, fp32) += sum_k A[i][k] B[k][j],  k = 0 .. 31:
//   A: lane l holds row i = l & 15, k = 8 (l >> 4) .. + 7;  B: lane l holds column j = l & 15, the same k;
//   D: lane l holds column j = l & 15, rows i = 4 (l >> 4) + r in register r = 0 .. 3.
// Layer 2 passes the WEIGHT rows as A and the activation rows as B: a lane owns ONE activation row (l & 15) and four
// output columns 4 (l >> 4) + r of the 16-column block.  Per accumulator and k-block (32 values): (w0, x1), (w1, x0), (w0, x0).
typedef float fx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fx4 mfma16(const h8 a, const h8 b, const fx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// CANONICAL partial logit of one (activation row, 16-column block): the lane's four columns in register order, then the four
// lane groups as (g0 + g1) + (g2 + g3) -- every lane of the row ends up with the same value.  bs / ws: bias and output-layer
// weights of the lane's four columns.
__device__ __forceinline__ float l2_partial16(const fx4 acc, const float descale, const fv4 bs, const fv4 ws) {
  float s = fmaxf(acc[0] * descale + bs[0], 0.0f) * ws[0];
  s += fmaxf(acc[1] * descale + bs[1], 0.0f) * ws[1];
  s += fmaxf(acc[2] * descale + bs[2], 0.0f) * ws[2];
  s += fmaxf(acc[3] * descale + bs[3], 0.0f) * ws[3];
  s = s + __shfl_xor(s, 16, 64);
  s = s + __shfl_xor(s, 32, 64);
  return s;
}

__device__ __forceinline__ void f16_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int BM, int BN, int BK, bool BLOCKS>
struct StageF16 {
  static constexpr int LDK = BK + 8;                // halves per LDS row
  static constexpr int CPR = BK / 8;                // 16-B chunks per row
  static constexpr int CA = BM * CPR / kBlock;      // chunks per thread per plane
  static constexpr int CB = BN * CPR / kBlock;
  static constexpr int RPP = kBlock / CPR;          // rows per pass
  h8 a[2][CA], b[2][CB];
  __device__ __forceinline__ void load(const GemmF16Args& g, int64_t m0, int n0, int kt, int tid) {
    const int kc = tid % CPR, r = tid / CPR;
    const _Float16* w = g.W + (int64_t)(n0 + r) * g.Kp + kt * BK + 8 * kc;
    const int64_t last = g.M - 1;  // rows past M re-read the last row; their results are never stored
    const _Float16* x = g.A + kt * BK + 8 * kc;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < CB; ++i) b[p][i] = *reinterpret_cast<const h8*>(w + p * g.plane_w + (int64_t)i * RPP * g.Kp);
    if (BLOCKS) {
      // block layout: row m, k-block kb = 64 halves [p0 x 32 | p1 x 32]; this thread's 8 values start at k = kt BK + 8 kc
      const int k0 = kt * BK + 8 * kc;
      const _Float16* blk = g.A + (k0 >> 5) * 64 + (k0 & 31);
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const int64_t m = m0 + r + RPP * i;
        const _Float16* src = blk + (m < last ? m : last) * (2 * g.lda);
        a[0][i] = *reinterpret_cast<const h8*>(src);
        a[1][i] = *reinterpret_cast<const h8*>(src + 32);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < CA; ++i) {
          const int64_t m = m0 + r + RPP * i;
          a[p][i] = *reinterpret_cast<const h8*>(x + p * g.plane_a + (m < last ? m : last) * g.lda);
        }
    }
  }
  // LDS: [A p0][A p1][B p0][B p1], each rows x LDK
  __device__ __forceinline__ void store(_Float16* s, int tid) const {
    const int kc = tid % CPR, r = tid / CPR;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int i = 0; i < CA; ++i) *reinterpret_cast<h8*>(&s[p * BM * LDK + (r + RPP * i) * LDK + 8 * kc]) = a[p][i];
#pragma unroll
      for (int i = 0; i < CB; ++i)
        *reinterpret_cast<h8*>(&s[2 * BM * LDK + p * BN * LDK + (r + RPP * i) * LDK + 8 * kc]) = b[p][i];
    }
  }
};

// MODE 0 reads its activations in block layout (the scaled input), MODE 1 as planes (H1, written by MODE 0).
// One register stage prefetched ahead (two stages and the no-load / no-barrier ablations were measured in round 1:
// profiles/r01_gemm_f16_variants.txt; neither is kept in the product kernel).
template <int TM, int TN, int BK, int MODE, int MINW>
__global__ __launch_bounds__(kBlock, MINW) void disc_gemm_f16_kernel(GemmF16Args g) {
  constexpr int BM = 64 * TM, BN = 64 * TN, LDK = BK + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* smem = reinterpret_cast<_Float16*>(smem_raw);
  int mt, nt;
  if (!f16_tile_of_block(g, mt, nt)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)mt * BM;
  const int n0 = nt * BN;
  const int nk = g.Kp / BK;

  fx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const _Float16* xa = smem + (wm * TM * 32 + li) * LDK + 8 * lh;                 // activation fragments, plane 0
  const _Float16* wb = smem + 2 * BM * LDK + (wn * TN * 32 + li) * LDK + 8 * lh;  // weight fragments, plane 0
  // hipcc orders the fragment reads itself (explicitly double-buffered fragment sets pinned with sched_barrier were
  // measured: +3 % on the bare ds_read + MFMA loop, spills at 3 workgroups / CU)
  const int ksteps = g.ksteps > 0 ? g.ksteps : g.Kp / 16;
  // MODE 1 (layer 2): v_mfma_f32_16x16x32_f16, one MFMA per k-block of 32 (see the top of this file); lane (i16, kq)
  fx4 c16[2 * TM][2 * TN];
#pragma unroll
  for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
    for (int b = 0; b < 2 * TN; ++b) c16[a][b] = fx4{0.0f, 0.0f, 0.0f, 0.0f};
  const int i16 = lane & 15, kq = lane >> 4;
  const _Float16* xa16 = smem + (wm * TM * 32 + i16) * LDK + 8 * kq;
  const _Float16* wb16 = smem + 2 * BM * LDK + (wn * TN * 32 + i16) * LDK + 8 * kq;
  // MODE 0 (layer 1): the MFMA of parity beta of a 32-column block takes the weight rows rho(beta, i) = 4 (i >> 1) + (i & 1) + 2 beta
  // (disc_gemm_f16_dma.hpp: after the epilogue's adjacent-lane exchange a lane holds four consecutive columns of a row)
  const _Float16* wb0[2] = {smem + 2 * BM * LDK + (wn * TN * 32 + 4 * (i16 >> 1) + (i16 & 1)) * LDK + 8 * kq,
                            smem + 2 * BM * LDK + (wn * TN * 32 + 4 * (i16 >> 1) + (i16 & 1) + 2) * LDK + 8 * kq};
  auto compute16 = [&](const int) {
    static_assert(BK % 32 == 0, "the 16 x 16 x 32 layers consume whole k-blocks of 32");
#pragma unroll
    for (int kb = 0; kb < BK / 32; ++kb) {
      h8 x0[2 * TM], x1[2 * TM], w0[2 * TN], w1[2 * TN];
#pragma unroll
      for (int a = 0; a < 2 * TM; ++a) {
        x0[a] = *reinterpret_cast<const h8*>(xa16 + a * 16 * LDK + 32 * kb);
        x1[a] = *reinterpret_cast<const h8*>(xa16 + BM * LDK + a * 16 * LDK + 32 * kb);
      }
#pragma unroll
      for (int b = 0; b < 2 * TN; ++b) {
        const _Float16* wsrc = MODE == 0 ? wb0[b & 1] + (b >> 1) * 32 * LDK : wb16 + b * 16 * LDK;
        w0[b] = *reinterpret_cast<const h8*>(wsrc + 32 * kb);
        w1[b] = *reinterpret_cast<const h8*>(wsrc + BN * LDK + 32 * kb);
      }
#pragma unroll
      for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
        for (int b = 0; b < 2 * TN; ++b) c16[a][b] = MODE == 1 ? mfma16(w0[b], x1[a], c16[a][b]) : mfma16(x1[a], w0[b], c16[a][b]);
#pragma unroll
      for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
        for (int b = 0; b < 2 * TN; ++b) c16[a][b] = MODE == 1 ? mfma16(w1[b], x0[a], c16[a][b]) : mfma16(x0[a], w1[b], c16[a][b]);
#pragma unroll
      for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
        for (int b = 0; b < 2 * TN; ++b) c16[a][b] = MODE == 1 ? mfma16(w0[b], x0[a], c16[a][b]) : mfma16(x0[a], w0[b], c16[a][b]);
    }
  };
  auto compute = [&](const int kt) {
    if constexpr (MODE == 0 || MODE == 1) { compute16(kt); return; }
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      if (kt * (BK / 16) + s >= ksteps) break;  // zero padding only (workgroup-uniform)
      h8 x0[TM], x1[TM], w0[TN], w1[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        x0[a] = *reinterpret_cast<const h8*>(xa + a * 32 * LDK + 16 * s);
        x1[a] = *reinterpret_cast<const h8*>(xa + BM * LDK + a * 32 * LDK + 16 * s);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        w0[b] = *reinterpret_cast<const h8*>(wb + b * 32 * LDK + 16 * s);
        w1[b] = *reinterpret_cast<const h8*>(wb + BN * LDK + b * 32 * LDK + 16 * s);
      }
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
    }
  };
  {
    StageF16<BM, BN, BK, MODE == 0> stg;
    stg.load(g, m0, n0, 0, tid);
    stg.store(smem, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) stg.load(g, m0, n0, kt + 1, tid);  // in flight under this tile's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      compute(kt);
      f16_lds_barrier();
      if (kt + 1 < nk) {
        stg.store(smem, tid);
        f16_lds_barrier();
      }
    }
  }
  __syncthreads();

  // ---- epilogue: register r of lane half lh is output column (r & 3) + 8 (r >> 2) + 4 lh of the block, lane li is
  //      activation row li
  const LayerScales sc = layer_scales(g.range, g.amax, g.layer);
  const float descale = sc.descale;
  const fv4* bias4 = reinterpret_cast<const fv4*>(g.bias + n0 + wn * (TN * 32) + 4 * lh);
  if (MODE == 0) {
    // relu(. + bias) -> planes of s_h H (planar: plane p at H + p * plane_h, rows of ldh halves).  Register r of lane (i16, kq)
    // is row 4 kq + r of the 16-row block and column rho(beta, i16) of the 32-column block (beta = MFMA parity): the same
    // arithmetic and adjacent-lane exchange as the LDS-DMA kernel's layer-1 epilogue -- even lanes end up with the p0 halves of
    // four consecutive columns 2 i .. 2 i + 3, odd lanes with the p1 halves of 2 (i - 1) .. + 3 -- stored as 8 bytes into the
    // lane's plane: 64 contiguous bytes per row, plane and 32-column block.
    const float s_h = sc.s_out, ds = descale * s_h;
    const uint32_t sel = (i16 & 1) ? 0x03020706u : 0x05040100u;  // v_perm_b32(neighbour, own, sel): odd [nb.hi, own.hi], even [own.lo, nb.lo]
    const int colw = n0 + wn * (TN * 32);
    float bsv[TN][2];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int be = 0; be < 2; ++be) bsv[b][be] = g.bias[colw + b * 32 + 4 * (i16 >> 1) + (i16 & 1) + 2 * be] * s_h;
    _Float16* const hp = g.H + (i16 & 1) * g.plane_h + colw + 4 * (i16 >> 1);  // this lane's plane, its first of four columns
    typedef uint32_t uw2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t grow = m0 + wm * (TM * 32) + a * 16 + 4 * kq + r;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          uw2 words;
#pragma unroll
          for (int be = 0; be < 2; ++be) {
            const float v = fmaxf(__builtin_fmaf(c16[a][2 * b + be][r], ds, bsv[b][be]), 0.0f);
            uint32_t own;
            asm("v_cvt_f16_f32 %0, %1" : "=v"(own) : "v"(v));
            asm("v_fma_mixhi_f16 %0, -%0, 1.0, %1 op_sel_hi:[1,0,0]" : "+v"(own) : "v"(v));
            const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
            words[be] = __builtin_amdgcn_perm(nb, own, sel);
          }
          if (grow < g.M) *reinterpret_cast<uw2*>(hp + grow * g.ldh + b * 32) = words;
        }
      }
  } else {
    // CANONICAL partial logits: one value per (row, 32-column block) = the sum of its two 16-column blocks' l2_partial16
    // (top of this file), written to partial[row][block] with `n_blocks` = N / 32 blocks per row.  Every f16 kernel,
    // whatever its tile, produces the same 16 values per row bit for bit (the accumulators are identical: same MFMA shape,
    // same k order, same three products per k-block), and disc_finalize_kernel adds them in one fixed tree -- so a row's logit
    // does not depend on the tile shape the shard size selected (shard equivalence, tests/test_gpu_shard_equivalence.py).
    float* red = reinterpret_cast<float*>(smem);  // [2 TN][BM]
    const fv4* b16 = reinterpret_cast<const fv4*>(g.bias + n0 + wn * (TN * 32) + 4 * kq);
    const fv4* w16 = reinterpret_cast<const fv4*>(g.w3 + n0 + wn * (TN * 32) + 4 * kq);
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const fv4 bs0 = b16[(2 * b) * 4], ws0 = w16[(2 * b) * 4], bs1 = b16[(2 * b + 1) * 4], ws1 = w16[(2 * b + 1) * 4];
#pragma unroll
      for (int a = 0; a < 2 * TM; ++a) {
        const float v = l2_partial16(c16[a][2 * b], descale, bs0, ws0) + l2_partial16(c16[a][2 * b + 1], descale, bs1, ws1);
        if (kq == 0) red[(wn * TN + b) * BM + wm * (TM * 32) + a * 16 + i16] = v;
      }
    }
    __syncthreads();
    constexpr int BPT = 2 * TN;  // 32-column blocks per tile
    const int n_blocks = g.N >> 5;
    for (int e = tid; e < BM * BPT; e += kBlock) {
      const int r = e / BPT, j = e - r * BPT;
      const int64_t row = m0 + r;
      if (row < g.M) g.partial[row * n_blocks + nt * BPT + j] = red[j * BM + r];
    }
  }
}

template <int TM, int TN, int BK>
constexpr int gemm_f16_lds_bytes() {
  constexpr int stage = 2 * (64 * TM + 64 * TN) * (BK + 8) * 2;
  constexpr int ep = 4 * 2 * 32 * (TN * 32 + 8) * 2;
  constexpr int red = 2 * TN * 64 * TM * 4;  // MODE 1: [2 TN][BM] floats
  return (stage > ep ? stage : ep) > red ? (stage > ep ? stage : ep) : red;
}

}  // namespace amp
