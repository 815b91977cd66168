// EXPERIMENT (not built into libamp_engine.so): weight-stationary, activation-streaming layer 1.  Correct (2.1e-7 of fp64 on
// tools/gemm_f16_bench.hip's check) but no faster than disc_gemm_f16_dma_kernel<0>: 57-58 us per 32 768-row launch against
// 54-55 us (a 10-wave variant with dedicated fill waves and one accumulator: 52 us).  See DESIGN.md section 7c.
// Layer 1 of the fp16-split discriminator forward for large shards with a short contraction (K*D = 166 -> six
// k-blocks), rebuilt for what bounds it.  With a 256 x 256 output tile per workgroup (disc_gemm_f16_dma_kernel<0>) a tile
// is a prologue, six k-blocks and a 256 KB epilogue: latency-bound (28 us per tile, 9 us of them MFMAs), and a second
// workgroup per CU (256 x 128 tiles, 80 KB of LDS: measured, same time) only shares that latency out.  So the tile
// structure is dropped: the WEIGHTS stay, the activations stream.
//
//   * a workgroup (8 waves) owns 256 output columns for the whole launch; wave w keeps the two fp16 planes of its 32
//     columns x 192 k of W1 in REGISTERS (24 x h8 = 96 VGPRs) -- no LDS traffic and no refills for W;
//   * it walks down the rows in steps of 32: one step = a 24-KB activation tile [6 k-blocks][32 rows][128 B] in LDS
//     (three stages), 12 k-steps x 3 MFMAs per wave into THREE 32 x 32 accumulators, one per product kind (w0 x0,
//     w0 x1, w1 x0: a single accumulator would be a chain of 36 dependent MFMAs, bound by their latency -- measured: 3x
//     slower than the matrix pipe allows), summed as acc0 + (acc1 + acc2); bias + ReLU + plane split; a wave-private
//     LDS slab; 16-B stores: a wave writes one full 128-B line per row ([p0 | p1] of its k-block of the hidden layer)
//     per step.  Stores leave continuously, under the MFMAs of the next step, and nobody ever waits for them;
//   * on gfx9 stores and LDS-DMA fills share vmcnt and complete out of order with each other, so a wave that stores
//     cannot count its fills.  The duties are split: waves 0-3 issue the fills (6 pieces per tile each; their vmcnt
//     holds fills only) and never store -- their slabs are double-buffered and stored one step later by waves 4-7, which
//     never touch vmcnt once their weights have arrived;
//   * one workgroup barrier per step: before barrier i waves 0-3 have waited for tile i; after it they issue tile i + 2
//     into the stage tile i - 1 was read from (every wave has retired those reads in front of barrier i), and wave 4 + w
//     stores the slab wave w wrote in step i - 1.
// Same block layout, XOR swizzle and epilogue arithmetic as the tile kernels; the sum of the three product kinds is
// taken once at the end instead of k-step by k-step (fp32 rounding differences only).
#pragma once
#include "disc_gemm_f16_dma.hpp"

namespace amp {

constexpr int kL1Waves = 8;                              // 32 columns each
constexpr int kL1Threads = kL1Waves * kWave;
constexpr int kL1Cols = 32 * kL1Waves;                   // columns per workgroup
constexpr int kL1Rows = 32;                              // rows per step
constexpr int kL1NQ = 6;                                 // k-blocks: the kernel is built for Kp = 192
constexpr int kL1Stages = 3;
constexpr int kL1Stage = kL1NQ * kL1Rows * 128;          // 24 KB
constexpr int kL1Slab = 2 * 32 * 40 * 2;                 // two planes x 32 rows x (32 + 8) halves
constexpr int kL1SlabsOff = kL1Stages * kL1Stage;        // slabs: waves 0-3 two each (steps alternate), waves 4-7 one each
constexpr int kL1BiasOff = kL1SlabsOff + 12 * kL1Slab;
constexpr int kL1LdsBytes = kL1BiasOff + kL1Cols * 4;    // 134 KB

__global__ __launch_bounds__(kL1Threads) void disc_gemm_f16_l1_kernel(GemmF16Args g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  // ---- which columns, which rows: the workgroups of one XCD (blockIdx & 7) take all column tiles of the same row
  // tiles at about the same time, so an activation tile comes from HBM / the Infinity Cache once per XCD
  const int j = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int ct = j % g.n_tiles, lanes = per_xcd / g.n_tiles;
  const int first = (blockIdx.x & 7) * lanes + j / g.n_tiles, stride = 8 * lanes;
  const int n_iter = first < g.m_tiles ? (g.m_tiles - first + stride - 1) / stride : 0;
  const int n0 = ct * kL1Cols + 32 * wave;
  float* const s_bias = reinterpret_cast<float*>(lds + kL1BiasOff);  // bias * s_h (relu_split4)
  if (tid < kL1Cols) s_bias[tid] = g.bias[ct * kL1Cols + tid] * layer_scales(g.range, g.amax, g.layer).s_out;
  const bool filler = wave < 4;  // waves 0-3 fill and never store; waves 4-7 store (their own slab and wave - 4's)
  constexpr int EPL = 40;        // slab row (halves)
  auto slab_of = [&](int w, int step) -> _Float16* {  // waves 0-3: slabs 2 w + (step & 1); waves 4-7: slab 4 + w
    return reinterpret_cast<_Float16*>(lds + kL1SlabsOff + (w < 4 ? 2 * w + (step & 1) : 4 + w) * kL1Slab);
  };

  // ---- the weights of the wave's 32 columns, both planes, every k-step
  h8 w0[2 * kL1NQ], w1[2 * kL1NQ];
  {
    const _Float16* wp = g.W + (int64_t)(n0 + li) * (2 * (int64_t)g.Kp) + 8 * lh;
#pragma unroll
    for (int s = 0; s < 2 * kL1NQ; ++s) {
      w0[s] = *reinterpret_cast<const h8*>(wp + (s >> 1) * 64 + (s & 1) * 16);
      w1[s] = *reinterpret_cast<const h8*>(wp + (s >> 1) * 64 + (s & 1) * 16 + 32);
    }
  }
  // ---- fills (waves 0-3): tile = [6 k-blocks][32 rows][128 B]; a piece is 8 rows x 128 B of one k-block; wave w takes
  // row group w of every k-block.  lane l: row 8 w + (l >> 3), stored chunk (l & 7) = source chunk (l & 7) ^ ((row >> 1) & 7),
  // and (row >> 1) & 7 = (4 w + (l >> 4)) & 7
  const int64_t last = g.M - 1;  // rows past M re-read the last row; their results are never stored
  const int64_t a_pitch = 2 * g.lda;
  const int fr = wave * 8 + (lane >> 3), fc = (lane & 7) ^ (lane >> 4) ^ ((wave & 1) << 2);
  auto fill = [&](int it) {
    const int64_t t0 = (first + (int64_t)it * stride) * kL1Rows;
    const int64_t m = t0 + fr < last ? t0 + fr : last;
    const _Float16* src = g.A + m * a_pitch + 8 * fc;
    unsigned char* sb = lds + (it % kL1Stages) * kL1Stage + wave * (8 * 128);
#pragma unroll
    for (int kb = 0; kb < kL1NQ; ++kb)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + kb * 64), (lptr_t)(sb + kb * (32 * 128)), 16, 0, 0);
  };
  if (filler) {
    // the weights were requested first: in-order return makes "at most the fills outstanding" mean "weights arrived"
    if (n_iter > 0) fill(0);
    if (n_iter > 1) fill(1);
  }
  const LayerScales sc = layer_scales(g.range, g.amax, g.layer);
  const float descale = sc.descale, s_h = sc.s_out;
  const int swz = (li >> 1) & 7;
  const int64_t h_pitch = 2 * g.ldh;
  // write-out of a slab: 8 chunks of 16 B per row in memory order (plane pl = ch >> 2, quarter ch & 3); lane l: row
  // (l >> 3) + 8 i, chunk l & 7
  auto store_slab = [&](const _Float16* ep, int it, int cols0) {
    const int64_t t0 = (first + (int64_t)it * stride) * kL1Rows;
    _Float16* const hrow = g.H + (t0 + (lane >> 3)) * h_pitch + (int64_t)(cols0 >> 5) * 64 + 8 * (lane & 7);
    const _Float16* const erow = ep + ((lane >> 2) & 1) * (32 * EPL) + (lane >> 3) * EPL + 8 * (lane & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const h8 v = *reinterpret_cast<const h8*>(erow + 8 * i * EPL);
      if (t0 + (lane >> 3) + 8 * i < g.M) *reinterpret_cast<h8*>(hrow + 8 * i * h_pitch) = v;
    }
  };
  __syncthreads();  // the bias is in LDS

  for (int it = 0; it <= n_iter; ++it) {  // one extra turn: waves 4-7 store the last slabs of waves 0-3
    if (filler && it < n_iter) {
      // tile `it` has landed when only the tile issued after it is outstanding (6 pieces)
      if (it + 1 < n_iter) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (filler && it + 2 < n_iter) fill(it + 2);  // into the stage of tile it - 1
    if (!filler && it > 0) store_slab(slab_of(wave - 4, it - 1), it - 1, n0 - 128);
    if (it == n_iter) break;
    const unsigned char* sa = lds + (it % kL1Stages) * kL1Stage + li * 128;
    fx16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 2 * kL1NQ; ++s) {
      // (p0, p1) pairs: the lane's eight values of k-step s are chunks 4 (s & 1) + 2 lh and + 1 of k-block s >> 1
      const unsigned char* row = sa + (s >> 1) * (32 * 128);
      const uv4 lo = *reinterpret_cast<const uv4*>(row + ((4 * (s & 1) + 2 * lh) ^ swz) * 16);
      const uv4 hi = *reinterpret_cast<const uv4*>(row + ((4 * (s & 1) + 2 * lh + 1) ^ swz) * 16);
      uv4 q0, q1;
      q0[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x05040100u); q1[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x07060302u);
      q0[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x05040100u); q1[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302u);
      q0[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x05040100u); q1[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x07060302u);
      q0[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x05040100u); q1[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302u);
      const h8 x0 = __builtin_bit_cast(h8, q0), x1 = __builtin_bit_cast(h8, q1);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[s], x1, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[s], x0, acc2, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[s], x0, acc0, 0, 0, 0);
    }
    // ---- epilogue of the step: register r of lane half lh is output column (r & 3) + 8 (r >> 2) + 4 lh, lane li is row
    // li; relu(. + bias) -> the two planes of s_h H into the wave's slab: one 128-B line [p0 | p1] per row
    _Float16* const ep = slab_of(wave, it);
#pragma unroll
    for (int grp4 = 0; grp4 < 4; ++grp4) {
      const fv4 bs = *reinterpret_cast<const fv4*>(&s_bias[32 * wave + 8 * grp4 + 4 * lh]);  // bias * s_h
      fv4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc0[4 * grp4 + i] + (acc1[4 * grp4 + i] + acc2[4 * grp4 + i]);
      h4 p0, p1;
      relu_split4(v, descale * s_h, bs, p0, p1);
      const int col = 8 * grp4 + 4 * lh;
      *reinterpret_cast<h4*>(&ep[li * EPL + col]) = p0;
      *reinterpret_cast<h4*>(&ep[32 * EPL + li * EPL + col]) = p1;
    }
    if (!filler) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      store_slab(ep, it, n0);
    }
    // every LDS access of this step (stage, slabs) retires before the next barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

}  // namespace amp
