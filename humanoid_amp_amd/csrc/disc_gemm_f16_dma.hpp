// GEMMs of the fp16-split discriminator forward for large shards: the same arithmetic, in the same order, as
// disc_gemm_f16_kernel (three v_mfma_f32_32x32x16_f16 per k-step into one fp32 accumulator, transposed accumulator
// tile; MODE 0: bias + ReLU -> fp16 planes of the hidden layer, activations read as (p0, p1) pairs; MODE 1: bias +
// ReLU + dot(w3) -> partial logits, activations read as planes), rebuilt around what bounds them on MI355X: the
// L2 -> CU fill rate (a 128 x 128 tile re-reads 2.1 GB of operands per 65 536-row launch, ~70 GB/s per CU at full
// MFMA rate = the measured L2-served per-CU ceiling) and the matrix pipe's idle time while its waves copy operands.
//
//   * 256 x 256 workgroup tile (half the operand bytes per MFMA of 128 x 128), 512 threads = 8 waves as 2 (rows) x 4
//     (columns), each wave 128 x 64 = 4 x 2 accumulator blocks; one workgroup per CU, two waves per SIMD.
//   * k-step = 16 = one stage: planes {A p0, A p1, B p0, B p1} of 256 rows x 32 B (MODE 0: the A half is 256 rows x
//     64 B of pairs, split into planes by v_perm_b32 after the fragment read) = 4 x 8 KB = 32 KB; four stages
//     (128 KB) filled by LDS-DMA (`global_load_lds_dwordx4`: 1 KiB per wave-instruction, no VGPRs, no ds_write):
//     32 pieces per stage, 4 per wave.  The pieces of k-step p + 3 are issued in k-step p, so they have two full
//     k-steps to land; the only memory wait in the loop is a COUNTED `s_waitcnt vmcnt(8)` (k-step p + 1 complete,
//     p + 2 and p + 3 in flight).
//   * a piece lands lane-linearly (wave-uniform base + 16 B x lane), so rows cannot be padded: the 16-B chunk c of
//     row r is stored at chunk c ^ ((r >> 3) & 1) -- applied to the per-lane SOURCE address of the fill and to the
//     fragment reads alike -- which spreads the 16 lanes of a ds_read_b128 group over the 16 slots of a bank row.
//   * every k-step is a LOAD segment (12 ds_read_b128 of fragments, the 4 pieces, the waits) followed by a MATRIX
//     segment (24 MFMAs at s_setprio 1) bracketed by two barriers; the two wave groups (waves 0-3 / 4-7: one wave of
//     each per SIMD) run the sequence one barrier apart, so a SIMD's matrix pipe alternates between its two waves.
//
// Ordering (MI355X_MICROARCH.md, LDS-DMA): a piece is visible to another wave's ds_read only after the issuing wave's
// counted vmcnt AND a barrier both have passed: the vmcnt(8) in the load segment of k-step p retires k-step p + 1,
// which is first read two barriers later.  A stage is refilled only by waves that have passed a barrier behind the
// lgkmcnt(0) that retired every wave's last fragment read of it (k-step p refills the stage of k-step p - 1).
#pragma once
#include "disc_gemm_f16.hpp"

namespace amp {

constexpr int kDmaThreads = 512, kDmaBM = 256, kDmaBN = 256, kDmaBK = 16, kDmaStages = 4;
constexpr int kDmaPlane = 256 * kDmaBK * 2;                 // 8 KB: one plane of one operand, 32-B rows
constexpr int kDmaStageBytes = 4 * kDmaPlane;               // 32 KB
constexpr int kDmaLdsBytes = kDmaStages * kDmaStageBytes;   // 128 KB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// XP != 0: ablations for tools/gemm_f16_bench.hip (wrong results): 1 = no fills in the loop, 2 = also no fragment reads,
// 3 = MODE 0 without the global stores of the epilogue, 4 = MODE 0 epilogue only (no k-loop)
template <int MODE, int XP = 0>
__global__ __launch_bounds__(kDmaThreads, 1) void disc_gemm_f16_dma_kernel(GemmF16Args g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  int mt, nt;
  if (!f16_tile_of_block(g, mt, nt)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)mt * kDmaBM;
  const int n0 = nt * kDmaBN;
  const int nk = XP == 4 ? 1 : g.Kp / kDmaBK;

  // ---- fill plan.  Weights (and MODE 1 activations): a piece is 32 rows x 32 B; wave w owns rows [32 w, 32 w + 32)
  // of each plane; lane l: row 32 w + (l >> 1), stored chunk (l & 1) = source chunk (l & 1) ^ ((row >> 3) & 1).
  // MODE 0 activations ((p0, p1) pairs, 64-B rows): a piece is 16 rows x 64 B; wave w owns pieces 2 w and 2 w + 1;
  // lane l: row 16 piece + (l >> 2), stored chunk (l & 3) = source chunk (l & 3) ^ ((row >> 2) & 3).
  const _Float16* src[4];
  {
    const int r = wave * 32 + (lane >> 1);
    const int c = (lane & 1) ^ ((r >> 3) & 1);
    src[2] = g.W + (int64_t)(n0 + r) * g.Kp + 8 * c;
    src[3] = src[2] + g.plane_w;
    const int64_t last = g.M - 1;  // rows past M re-read the last row; their results are never stored
    if (MODE == 1) {
      const int64_t m = m0 + r < last ? m0 + r : last;
      src[0] = g.A + m * g.lda + 8 * c;
      src[1] = src[0] + g.plane_a;
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rp = wave * 32 + j * 16 + (lane >> 2);
        const int cp = (lane & 3) ^ ((rp >> 2) & 3);
        const int64_t m = m0 + rp < last ? m0 + rp : last;
        src[j] = g.A + 2 * (m * g.lda + 4 * cp);  // lda counts pairs; 16 B = 4 pairs
      }
    }
  }
  constexpr int kAStep = MODE == 1 ? kDmaBK : 2 * kDmaBK;  // halves per k-step along an activation row
  auto fill = [&](int p, int stage) {  // the wave's four pieces of k-step p
    unsigned char* sb = lds + stage * kDmaStageBytes;
    if (MODE == 1) {
      __builtin_amdgcn_global_load_lds((gptr_t)(src[0] + p * kAStep), (lptr_t)(sb + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src[1] + p * kAStep), (lptr_t)(sb + kDmaPlane + wave * 1024), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((gptr_t)(src[0] + p * kAStep), (lptr_t)(sb + wave * 2048), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src[1] + p * kAStep), (lptr_t)(sb + wave * 2048 + 1024), 16, 0, 0);
    }
    __builtin_amdgcn_global_load_lds((gptr_t)(src[2] + p * kDmaBK), (lptr_t)(sb + 2 * kDmaPlane + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(src[3] + p * kDmaBK), (lptr_t)(sb + 3 * kDmaPlane + wave * 1024), 16, 0, 0);
  };

  // ---- fragment addresses (bytes inside a stage).  32-B rows: row r, lane half lh -> chunk lh ^ ((r >> 3) & 1); the
  // wave's rows start at multiples of 64, so (r >> 3) & 1 = (li >> 3) & 1.  64-B pair rows: the lane's eight elements
  // are chunks 2 lh and 2 lh + 1, each ^ ((r >> 2) & 3).
  const int ca = (lh ^ ((li >> 3) & 1)) * 16;
  const int brow = 2 * kDmaPlane + (wn * 64 + li) * 32 + ca;
  const int arow = MODE == 1 ? (wm * 128 + li) * 32 + ca : (wm * 128 + li) * 64 + (((2 * lh) ^ ((li >> 2) & 3)) * 16);
  const int arow2 = (wm * 128 + li) * 64 + (((2 * lh + 1) ^ ((li >> 2) & 3)) * 16);  // MODE 0: the second chunk

  fx16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  h8 x0[4], x1[4], w0[2], w1[2];
  auto read_frags = [&](const unsigned char* sb) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (MODE == 1) {
        x0[a] = *reinterpret_cast<const h8*>(sb + arow + a * 32 * 32);
        x1[a] = *reinterpret_cast<const h8*>(sb + kDmaPlane + arow + a * 32 * 32);
      } else {
        const uv4 lo = *reinterpret_cast<const uv4*>(sb + arow + a * 32 * 64), hi = *reinterpret_cast<const uv4*>(sb + arow2 + a * 32 * 64);
        uv4 q0, q1;
        q0[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x05040100u); q1[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x07060302u);
        q0[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x05040100u); q1[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302u);
        q0[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x05040100u); q1[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x07060302u);
        q0[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x05040100u); q1[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302u);
        x0[a] = __builtin_bit_cast(h8, q0);
        x1[a] = __builtin_bit_cast(h8, q1);
      }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      w0[b] = *reinterpret_cast<const h8*>(sb + brow + b * 32 * 32);
      w1[b] = *reinterpret_cast<const h8*>(sb + kDmaPlane + brow + b * 32 * 32);
    }
  };

  // Two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run the same sequence one barrier apart: while one
  // group's waves are in a matrix segment, their SIMD partners are in a load segment.
  const int grp = wave >> 2;
  fill(0, 0);
  if (nk > 1) fill(1, 1);
  if (nk > 2) fill(2, 2);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // k-step 0 is visible to every wave
  if (grp == 1) __builtin_amdgcn_s_barrier();
  int stage = 0;
  for (int p = 0; p < nk; ++p) {
    // ---- load segment: fragments of k-step p; pieces of k-step p + 3 into the stage k-step p - 1 was read from
    // (every read of it was retired by an lgkmcnt(0) in front of a barrier this wave has passed); then this wave's
    // pieces of k-step p + 1 must have landed
    if (XP < 2 || p == 0) read_frags(lds + stage * kDmaStageBytes);
    if (XP == 0 && p + 3 < nk) fill(p + 3, stage == 0 ? 3 : stage - 1);
    if (p + 3 < nk) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (p + 2 < nk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // ---- matrix segment: nothing but the 24 MFMAs, at raised priority, between two barriers
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[b], x1[a], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[b], x0[a], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[b], x0[a], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stage = stage == 3 ? 0 : stage + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  __syncthreads();  // every wave is done with the stages: the reduction scratch below reuses stage 0

  // ---- epilogue: register r of lane half lh is output column (r & 3) + 8 (r >> 2) + 4 lh of the 32-wide block,
  //      lane li is activation row li
  const LayerScales sc = layer_scales(g.range, g.amax, g.layer);
  const float descale = sc.descale;
  const fv4* bias4 = reinterpret_cast<const fv4*>(g.bias + n0 + wn * 64 + 4 * lh);
  if (MODE == 0) {
    // relu(. + bias) -> planes of s_h H, transposed through a wave-private LDS slab per 32 rows so that a lane stores
    // 16 B and eight lanes cover one 128-B row segment of a plane (as disc_gemm_f16_kernel MODE 0)
    constexpr int EPL = 64 + 8;  // padded slab row (halves)
    const float s_h = sc.s_out;
    _Float16* ep = reinterpret_cast<_Float16*>(lds) + wave * (2 * 32 * EPL);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int grp4 = 0; grp4 < 4; ++grp4) {
          const fv4 bs = bias4[b * 8 + grp4 * 2];
          fv4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(acc[a][b][4 * grp4 + i] * descale + bs[i], 0.0f) * s_h;
          h4 p0, p1;
          split_planes4(v, p0, p1);
          const int col = b * 32 + 8 * grp4 + 4 * lh;
          *reinterpret_cast<h4*>(&ep[li * EPL + col]) = p0;
          *reinterpret_cast<h4*>(&ep[32 * EPL + li * EPL + col]) = p1;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = lane + 64 * i, pl = idx >> 8, row = (idx >> 3) & 31, q = idx & 7;
        const h8 v = *reinterpret_cast<const h8*>(&ep[pl * 32 * EPL + row * EPL + 8 * q]);
        const int64_t grow = m0 + wm * 128 + a * 32 + row;
        if (grow < g.M && (XP != 3 || v[0] == (_Float16)12345.0f))
          *reinterpret_cast<h8*>(&g.H[pl * g.plane_h + grow * g.ldh + n0 + wn * 64 + 8 * q]) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    return;
  }
  const fv4* w34 = reinterpret_cast<const fv4*>(g.w3 + n0 + wn * 64 + 4 * lh);
  float* red = reinterpret_cast<float*>(lds);  // [4][256]
  float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int grp4 = 0; grp4 < 4; ++grp4) {
      const fv4 bs = bias4[b * 8 + grp4 * 2], ws = w34[b * 8 + grp4 * 2];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[a] += fmaxf(acc[a][b][4 * grp4 + i] * descale + bs[i], 0.0f) * ws[i];
    }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float v = sum[a] + __shfl_xor(sum[a], 32, 64);  // the other lane half holds the other columns
    if (lh == 0) red[wn * kDmaBM + wm * 128 + a * 32 + li] = v;
  }
  __syncthreads();
  if (tid < kDmaBM) {
    const int64_t row = m0 + tid;
    // fixed order: (columns 0-63 + 64-127) + (128-191 + 192-255)
    if (row < g.M)
      g.partial[row * g.n_tiles + nt] = (red[tid] + red[kDmaBM + tid]) + (red[2 * kDmaBM + tid] + red[3 * kDmaBM + tid]);
  }
}

}  // namespace amp
