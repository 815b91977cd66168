// Split-precision variant of the discriminator GEMM: fp32 operands are carried as P bf16 planes
//   x = x1 + x2 (+ x3),   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (residuals exact in fp32)
// and the product is the sum of the plane pairs of "order" <= P - 1 on v_mfma_f32_32x32x16_bf16 (fp32 accumulate):
//   P = 3 ("bf16x6"): x1w1 + x1w2 + x2w1 + x1w3 + x2w2 + x3w1   -- dropped terms <= 2^-24 |x w|: fp32-level accuracy
//   P = 2 ("bf16x3"): x1w1 + x1w2 + x2w1                          -- dropped terms ~ 2^-16 |x w|
// bf16 MFMA runs at 16x the fp32-MFMA rate on gfx950, so 6 (3) of them per k-step are 2.7x (5.3x) faster than the
// native fp32 path of disc_gemm.hpp.  OPT-IN (amp_disc_set_precision): the default path stays native fp32.
//
// Layout: planes are [P][rows][Kp] bf16 in HBM (Kp % 16 == 0), tiles [P][rows][16 + 8 pad] bf16 in LDS.  For
// mfma_f32_32x32x16_bf16 lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] and B[k = 8h + j][col r], j = 0..7:
// one 16-B LDS read per fragment for both operands, conflict-free with the 48-B row pitch.  C/D layout as the f32 MFMA.
// Tile 128 x 128 x 16 (or 64 x 64 x 16), 4 waves 2 x 2, one LDS stage, register-prefetched staging (see disc_gemm.hpp).
#pragma once
#include "disc_gemm.hpp"

namespace amp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct SplitGemmArgs {
  const __bf16* A; int64_t a_plane; int64_t lda; int64_t M;   // A planes [P][M][lda]
  const __bf16* W; int64_t w_plane; int32_t Kp;               // W planes [P][N][Kp]
  const float* bias; int32_t N;
  __bf16* C; int64_t c_plane; int64_t ldc;                    // mode 0: output planes [P][M][ldc]
  const float* w3; float* partial; int32_t n_tiles;           // mode 1: partial [M][n_tiles]
  int32_t m_tiles;
};

__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

template <int BM_, int BN_, int P, int MODE, int MINW_>
__global__ __launch_bounds__(kBlock, MINW_) void disc_gemm_split_kernel(SplitGemmArgs g) {
  constexpr int TM = BM_ / 64, TN = BN_ / 64;
  constexpr int PITCH = 24;                                   // bf16 per LDS row: 16 data + 8 pad (48 B)
  constexpr int A_ELEMS = P * BM_ * PITCH, B_ELEMS = P * BN_ * PITCH;
  constexpr int EP_BYTES = 4 * 32 * (TN * 32 + 4) * 4;        // mode-0 epilogue transpose region (fp32)
  constexpr int SMEM_BYTES = (A_ELEMS + B_ELEMS) * 2 > EP_BYTES ? (A_ELEMS + B_ELEMS) * 2 : EP_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* Bs = As + A_ELEMS;

  GemmArgs map{};  // reuse the XCD-aware renumbering
  map.m_tiles = g.m_tiles;
  map.n_tiles = g.n_tiles;
  int mt, nt;
  if (!tile_of_block(map, mt, nt)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)mt * BM_;
  const int n0 = nt * BN_;
  const int nk = g.Kp / 16;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // staging: a plane tile is rows x 2 pieces of 16 B; piece p -> (plane, row, half)
  constexpr int A_PIECES = P * BM_ * 2, B_PIECES = P * BN_ * 2;
  constexpr int A_PER = (A_PIECES + kBlock - 1) / kBlock, B_PER = (B_PIECES + kBlock - 1) / kBlock;
  f4 ra[A_PER], rb[B_PER];
  const int64_t last = g.M - 1;
  auto load = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int p = tid + i * kBlock;
      if (A_PIECES % kBlock == 0 || p < A_PIECES) {
        const int plane = p / (BM_ * 2), rem = p % (BM_ * 2), row = rem >> 1, half = rem & 1;
        int64_t m = m0 + row;
        m = m < last ? m : last;
        ra[i] = *reinterpret_cast<const f4*>(g.A + plane * g.a_plane + m * g.lda + kt * 16 + half * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int p = tid + i * kBlock;
      if (B_PIECES % kBlock == 0 || p < B_PIECES) {
        const int plane = p / (BN_ * 2), rem = p % (BN_ * 2), row = rem >> 1, half = rem & 1;
        rb[i] = *reinterpret_cast<const f4*>(g.W + plane * g.w_plane + (int64_t)(n0 + row) * g.Kp + kt * 16 + half * 8);
      }
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int p = tid + i * kBlock;
      if (A_PIECES % kBlock == 0 || p < A_PIECES) {
        const int plane = p / (BM_ * 2), rem = p % (BM_ * 2), row = rem >> 1, half = rem & 1;
        *reinterpret_cast<f4*>(As + (plane * BM_ + row) * PITCH + half * 8) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int p = tid + i * kBlock;
      if (B_PIECES % kBlock == 0 || p < B_PIECES) {
        const int plane = p / (BN_ * 2), rem = p % (BN_ * 2), row = rem >> 1, half = rem & 1;
        *reinterpret_cast<f4*>(Bs + (plane * BN_ + row) * PITCH + half * 8) = rb[i];
      }
    }
  };

  const int arow = wm * (TM * 32) + li, brow = wn * (TN * 32) + li;
  load(0);
  store();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load(kt + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above the MFMAs (see disc_gemm.hpp)
    bf16x8 x[TM][P], y[TN][P];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int pl = 0; pl < P; ++pl)
        x[a][pl] = *reinterpret_cast<const bf16x8*>(As + (pl * BM_ + arow + a * 32) * PITCH + 8 * lh);
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int pl = 0; pl < P; ++pl)
        y[b][pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BN_ + brow + b * 32) * PITCH + 8 * lh);
    // plane pairs of order <= P - 1, smallest contributions first
#pragma unroll
    for (int order = P - 1; order >= 0; --order)
#pragma unroll
      for (int pa = 0; pa <= order; ++pa) {
        const int pw = order - pa;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = MODE == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[b][pw], x[a][pa], acc[a][b], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[a][pa], y[b][pw], acc[a][b], 0, 0, 0);
      }
    lds_barrier();
    if (kt + 1 < nk) {
      store();
      lds_barrier();
    }
  }
  __syncthreads();

  if (MODE == 0) {
    // bias + ReLU in fp32, transpose through LDS, split into planes, 8-B stores (4 columns x bf16) per plane
    constexpr int W = TN * 32, EPL = W + 4, QPR = W / 4;
    float* ep = reinterpret_cast<float*>(smem_raw) + wave * (32 * EPL);
    float bias[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) bias[b] = g.bias[n0 + wn * W + b * 32 + li];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
        for (int b = 0; b < TN; ++b) ep[row * EPL + b * 32 + li] = fmaxf(acc[a][b][r] + bias[b], 0.0f);
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < (32 * QPR) / 64; ++i) {
        const int idx = lane + 64 * i, row = idx / QPR, q = idx % QPR;
        const f4 v = *reinterpret_cast<const f4*>(&ep[row * EPL + 4 * q]);
        const int64_t grow = m0 + wm * (TM * 32) + a * 32 + row;
        if (grow < g.M) {
          bf16x4 o[3];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            __bf16 s0, s1, s2;
            split3(v[c], s0, s1, s2);
            o[0][c] = s0; o[1][c] = s1; o[2][c] = s2;
          }
#pragma unroll
          for (int pl = 0; pl < P; ++pl)
            *reinterpret_cast<bf16x4*>(g.C + pl * g.c_plane + grow * g.ldc + n0 + wn * W + 4 * q) = o[pl];
        }
      }
      __syncthreads();
    }
  } else {
    // transposed accumulators: the dot with w3 is a per-lane sum over registers (see disc_gemm.hpp)
    float* red = reinterpret_cast<float*>(smem_raw);  // [2][BM]
    const f4* bias4 = reinterpret_cast<const f4*>(g.bias + n0 + wn * (TN * 32) + 4 * lh);
    const f4* w34 = reinterpret_cast<const f4*>(g.w3 + n0 + wn * (TN * 32) + 4 * lh);
    float sum[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) sum[a] = 0.0f;
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        const f4 bs = bias4[b * 8 + grp * 2], ws = w34[b * 8 + grp * 2];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int i = 0; i < 4; ++i) sum[a] += fmaxf(acc[a][b][4 * grp + i] + bs[i], 0.0f) * ws[i];
      }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const float v = sum[a] + __shfl_xor(sum[a], 32, 64);
      if (lh == 0) red[wn * BM_ + wm * (TM * 32) + a * 32 + li] = v;
    }
    __syncthreads();
    if (tid < BM_) {
      const int64_t row = m0 + tid;
      if (row < g.M) g.partial[row * g.n_tiles + nt] = red[tid] + red[BM_ + tid];
    }
  }
}

// fp32 [rows, k] (row stride in floats) -> P bf16 planes [P][rows][kp], optional scaler, zero padding.
__global__ __launch_bounds__(kBlock) void disc_split_rows_kernel(const float* __restrict__ x, int64_t row_stride, int64_t rows,
                                                                 int k, int kp, const float* __restrict__ mean,
                                                                 const float* __restrict__ den, float clip, int planes,
                                                                 __bf16* __restrict__ out, int64_t plane_stride,
                                                                 const float* __restrict__ task, float* __restrict__ task_copy) {
  const int q_per_row = kp >> 2;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (task && e < rows) task_copy[e] = task[e];
  if (e >= rows * q_per_row) return;
  const int64_t m = e / q_per_row;
  const int c0 = (int)(e - m * q_per_row) * 4;
  const float* row = x + m * row_stride;
  bf16x4 o[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + i;
    float v = 0.0f;
    if (c < k) {
      v = row[c];
      if (mean) {
        v = (v - mean[c]) / den[c];
        v = fminf(fmaxf(v, -clip), clip);
      }
    }
    __bf16 s0, s1, s2;
    split3(v, s0, s1, s2);
    o[0][i] = s0; o[1][i] = s1; o[2][i] = s2;
  }
  for (int pl = 0; pl < planes; ++pl) *reinterpret_cast<bf16x4*>(out + pl * plane_stride + m * kp + c0) = o[pl];
}

}  // namespace amp
