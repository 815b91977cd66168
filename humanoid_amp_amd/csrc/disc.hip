// Discriminator style reward: scaler -> Linear(K*D,1024)+ReLU -> Linear(1024,512)+ReLU -> Linear(512,1)
// -> -log(max(1 - sigmoid, 1e-4)) * scale -> reward mix, at fp32 accuracy (the path's 1e-5 budget).
//
// Two GEMM engines share the layout below (amp_disc_set_precision):
//   AMP_DISC_F16X3 (default)  fp32 operands as two fp16 planes, three v_mfma_f32_32x32x16_f16 per k-step into one fp32
//                             accumulator (disc_gemm_f16.hpp): fp32-class error at 3/16 of the fp32 pipe's MFMA cycles.
//   AMP_DISC_FP32             v_mfma_f32_32x32x2_f32 on fp32 operands (disc_gemm.hpp): exact fp32 fma chain.
//
//   scale    one pass over amp_obs: RunningStandardScaler (exact fp32 divide, once per element) + zero padding of
//            K*D to the k-tile -> Xs in the workspace (fp32 rows, or the fp16 planes of s_x * Xs in block layout).
//   layer 1  GEMM [M, k] x [k, 1024]: bias + ReLU in the epilogue, transposed through LDS so every store is a full
//            row segment; H1 written once to the workspace (fp32, or the two planes of s_h * H1).
//   layer 2  GEMM [M,1024] x [1024,512]: bias + ReLU + the 512->1 output layer as a per-lane dot with w3 over the
//            transposed accumulator tile; only per-(row, column-tile) partial logits leave the kernel.
//   finalize fixed-order sum of the column-tile partials + b3, style reward, reward mix.
//
// Plane scales are powers of two derived from BOUNDS, not from the data: |Xs| <= clip (the scaler's clamp; without a
// clamp an abs-max pass over the scaled input supplies it), |W| <= max |W|, |H1| <= max_j sum_k |W1[j,k]| * bound_x +
// max |b1|.  A loose bound costs nothing (fp16 keeps 2^-11 relative precision over 2^30 below the bound), and results
// do not depend on which other rows share the batch.
#include <algorithm>

#include "disc_gemm.hpp"
#include "disc_gemm_f16.hpp"
#include "disc_gemm_f16_dma.hpp"
#include "disc_mlp_fused.hpp"
#include "compact_kernels.hpp"


typedef float f4 __attribute__((ext_vector_type(4)));  // native vector: HIP's f4 struct turns into memcpy -> scratch

using amp::DiscRange;  // device-resident range record of the fp16-split path (disc_gemm_f16.hpp)

struct AmpDisc {
  int32_t in_dim, h1, h2, k1p;
  float* w1p;  // [h1, k1p] zero padded along k
  float* b1;   // [h1]
  float* w2;   // [h2, h1]
  float* b2;   // [h2]
  float* w3;   // [h2]
  float* b3;   // [1]
  float* mean; // [k1p] fp32
  float* den;  // [k1p] sqrt(var) + eps
  float clip;
  bool has_scaler;
  int32_t mode;       // AMP_DISC_F16X3 / AMP_DISC_FP32
  int32_t k1h;        // in_dim padded to the fp16 k-tile
  _Float16* w1h;      // [2][h1][k1h] planes of s_w1 * W1
  _Float16* w2h;      // [2][h2][h1]  planes of s_w2 * W2
  _Float16* w1b;      // the same planes in block layout [h1][k1h / 32][2][32] (LDS-DMA kernels)
  _Float16* w2b;      // [h2][h1 / 32][2][32]
  DiscRange* range;   // device
  // per-handle plan overrides (amp_disc_set_plan; -1 = automatic: the process environment, then the measured defaults)
  int64_t ovr_fused_min_rows = -1;   // >= 128: the fused two-layer plan's threshold; INT64_MAX: plan off
};

namespace amp {

constexpr int kMinN = 128;  // h1 / h2 must be multiples of the widest column tile
constexpr int kPadK = 16;  // in_dim is zero-padded to a multiple of the fp32 layer-1 k-tile
constexpr int kPadKH = 32; // ... and of the fp16 layer-1 k-tile
constexpr int64_t kWsHeader = 256;  // workspace header: [0] abs-max of the scaled input (dynamic bound)

__device__ __forceinline__ void disc_finalize_body(const int64_t i, const float* __restrict__ partial, int n_tiles,
                                                   const float* __restrict__ b3, int64_t M, float scale,
                                                   const float* __restrict__ task, float task_w, float style_w,
                                                   float* __restrict__ logits, float* __restrict__ style,
                                                   float* __restrict__ combined) {
  if (i >= M) return;
  const float* p = partial + i * n_tiles;
  float s = 0.0f;
  if (n_tiles == 16) {  // the f16 engine's canonical 32-column block sums (h2 = 512): one fixed balanced tree
    const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4),
             c = *reinterpret_cast<const f4*>(p + 8), d = *reinterpret_cast<const f4*>(p + 12);
    s = (((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]))) +
        (((c[0] + c[1]) + (c[2] + c[3])) + ((d[0] + d[1]) + (d[2] + d[3])));
  } else if (n_tiles == 4) {
    s = (p[0] + p[1]) + (p[2] + p[3]);
  } else if (n_tiles == 8) {
    s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  } else {
    for (int t = 0; t < n_tiles; ++t) s += p[t];
  }
  const float lg = s + b3[0];
  // -log(max(1 - 1 / (1 + exp(-logit)), 1e-4)) * discriminator_reward_scale   (skrl AMP)
  const float pr = 1.0f - 1.0f / (1.0f + expf(-lg));
  const float st = -logf(fmaxf(pr, 0.0001f)) * scale;
  if (logits) logits[i] = lg;
  if (style) style[i] = st;
  if (combined) combined[i] = task ? task_w * task[i] + style_w * st : style_w * st;
}

__global__ __launch_bounds__(kBlock) void disc_finalize_kernel(const float* __restrict__ partial, int n_tiles,
                                                               const float* __restrict__ b3, int64_t M, float scale,
                                                               const float* __restrict__ task, float task_w,
                                                               float style_w, float* __restrict__ logits,
                                                               float* __restrict__ style, float* __restrict__ combined) {
  disc_finalize_body((int64_t)blockIdx.x * kBlock + threadIdx.x, partial, n_tiles, b3, M, scale, task, task_w, style_w, logits,
                     style, combined);
}

// The step's two latency-bound tail launches as ONE (horizontal fusion; they share nothing): workgroups
// [0, compact_blocks) run the reset-id compaction of amp_reset_compact_tiles, the rest the finalize.  Bit-identical
// to the separate launches by construction (the same device bodies).
struct CompactLaunch {
  const uint8_t* mask; const int32_t* counts; int64_t N, n_tiles, n_counts; int sub; int64_t* ids; int64_t* count; int blocks;
};
__global__ __launch_bounds__(kBlock) void step_tail_kernel(CompactLaunch c, const float* __restrict__ partial, int n_tiles,
                                                           const float* __restrict__ b3, int64_t M, float scale,
                                                           const float* __restrict__ task, float task_w, float style_w,
                                                           float* __restrict__ logits, float* __restrict__ style,
                                                           float* __restrict__ combined) {
  if ((int)blockIdx.x < c.blocks) {
    compact_scatter_body(blockIdx.x, c.mask, c.counts, c.N, c.n_tiles, c.sub, c.n_counts, c.ids, c.count);
    return;
  }
  disc_finalize_body((int64_t)(blockIdx.x - c.blocks) * kBlock + threadIdx.x, partial, n_tiles, b3, M, scale, task, task_w,
                     style_w, logits, style, combined);
}

__global__ void disc_scaler_kernel(const double* __restrict__ mean64, const double* __restrict__ var64, int n, int np,
                                   float eps, float* __restrict__ mean, float* __restrict__ den) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  if (i < n) {
    mean[i] = (float)mean64[i];
    den[i] = sqrtf((float)var64[i]) + eps;
  } else {
    mean[i] = 0.0f;
    den[i] = 1.0f;
  }
}

__global__ void disc_pad_rows_kernel(const float* __restrict__ src, int rows, int k, int kp, float* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)rows * kp) return;
  const int r = (int)(e / kp), c = (int)(e - (int64_t)r * kp);
  dst[e] = c < k ? src[(int64_t)r * k + c] : 0.0f;
}

// amp_obs [M, in] (any row stride) -> Xs [M, kp]: skrl RunningStandardScaler
//   clamp((x - mean) / (sqrt(var) + eps), -clip, clip)   (exact fp32 divide, once per element), zero padded to kp.
__global__ __launch_bounds__(kBlock) void disc_scale_pad_kernel(const float* __restrict__ x, int64_t row_stride, int64_t M,
                                                                int k, int kp, const float* __restrict__ mean,
                                                                const float* __restrict__ den, float clip,
                                                                float* __restrict__ xs, const float* __restrict__ task,
                                                                float* __restrict__ task_copy) {
  const int q_per_row = kp >> 2;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  // snapshot of the task reward: after this kernel the caller may overwrite amp_obs AND task_reward (next env step)
  if (task && e < M) task_copy[e] = task[e];
  if (e >= M * q_per_row) return;
  const int64_t m = e / q_per_row;
  const int c0 = (int)(e - m * q_per_row) * 4;
  const float* row = x + m * row_stride;
  f4 o;
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
    o[i] = v;
  }
  *reinterpret_cast<f4*>(xs + m * kp + c0) = o;
}

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---- fp16-split path: ranges, plane scales, input split ------------------------------------------------------------

// max |W1|, max_j sum_k |W1[j,k]|, max |b1|, max |W2| -> DiscRange (runs whenever the weights change, i.e. after every
// training step).  Non-negative floats order like their bit patterns, so the four maxima are atomicMax on uints
// (order-independent: deterministic) into the record's first four slots, zeroed beforehand; a one-thread kernel then
// turns max |W| into plane scales in place.  One wave per W1 row, grid-stride over W2.
__global__ __launch_bounds__(kBlock) void disc_weight_range_kernel(const float* __restrict__ w1p, int h1, int k1p,
                                                                   const float* __restrict__ b1, const float* __restrict__ w2,
                                                                   int64_t n2, DiscRange* __restrict__ out) {
  unsigned* slot = out->raw;  // [0] max |W1|, [1] max |W2|, [2] max row sum, [3] max |b1| (zero on entry: see DiscRange)
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  float wmax1 = 0.0f, rsum_max = 0.0f, bmax = 0.0f, wmax2 = 0.0f;
  for (int64_t j = wave; j < h1; j += n_waves) {
    float rs = 0.0f;
    for (int k = lane; k < k1p; k += 64) {
      const float a = fabsf(w1p[j * k1p + k]);
      wmax1 = fmaxf(wmax1, a);
      rs += a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rs += __shfl_xor(rs, o, 64);
    rsum_max = fmaxf(rsum_max, rs);
    if (lane == 0) bmax = fmaxf(bmax, fabsf(b1[j]));
  }
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n2; e += (int64_t)gridDim.x * kBlock) wmax2 = fmaxf(wmax2, fabsf(w2[e]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    wmax1 = fmaxf(wmax1, __shfl_xor(wmax1, o, 64));
    wmax2 = fmaxf(wmax2, __shfl_xor(wmax2, o, 64));
    bmax = fmaxf(bmax, __shfl_xor(bmax, o, 64));
  }
  // one atomic per workgroup and slot: atomics on one address serialise in the L2 (~50 ns each; 1 024 of them per slot
  // made this kernel 51 us)
  __shared__ float red[4][kBlock / kWave];
  if (lane == 0) {
    const int w = threadIdx.x >> 6;
    red[0][w] = wmax1; red[1][w] = wmax2; red[2][w] = rsum_max; red[3][w] = bmax;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float m = 0.0f;
    for (int w = 0; w < kBlock / kWave; ++w) m = fmaxf(m, red[threadIdx.x][w]);
    atomicMax(slot + threadIdx.x, __float_as_uint(m));
  }
  // The workgroup that takes the last ticket publishes the record (a one-thread kernel of its own, behind a memset of the slots,
  // until round 4: two more launches behind every training step).  Every workgroup's four atomics are performed before its
  // ticket (barrier + fence); the finisher takes the maxima out through atomic exchanges (device-coherent), which also leaves
  // the accumulators zero for the next launch.
  __shared__ unsigned s_ticket;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_ticket = atomicAdd(&out->ticket, 1u);
  }
  __syncthreads();
  if (s_ticket == gridDim.x - 1 && threadIdx.x == 0) {
    __threadfence();
    const float m1 = __uint_as_float(atomicExch(slot + 0, 0u)), m2 = __uint_as_float(atomicExch(slot + 1, 0u));
    const float rs = __uint_as_float(atomicExch(slot + 2, 0u)), bm = __uint_as_float(atomicExch(slot + 3, 0u));
    out->s_w1 = plane_scale(m1);
    out->s_w2 = plane_scale(m2);
    out->wsum1 = rs * 1.0001f;  // the row sums were rounded (and summed in lane order): keep the bound a bound
    out->bmax1 = bm;
    atomicExch(&out->ticket, 0u);
  }
}

__global__ void disc_set_clip_kernel(DiscRange* r, float clip) { r->clip = clip; }

// abs-max of the scaled input (only when no clamp bounds it): one atomicMax per workgroup on the bit pattern of a
// non-negative float (order-independent, hence deterministic)
__global__ __launch_bounds__(kBlock) void disc_absmax_kernel(const float* __restrict__ x, int64_t row_stride, int64_t M, int k,
                                                             const float* __restrict__ mean, const float* __restrict__ den,
                                                             float clip, unsigned* __restrict__ amax) {
  __shared__ float red[kBlock];
  float m = 0.0f;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < M * k; e += (int64_t)gridDim.x * kBlock) {
    const int64_t r = e / k;
    const int c = (int)(e - r * k);
    float v = x[r * row_stride + c];
    if (mean) v = fminf(fmaxf((v - mean[c]) / den[c], -clip), clip);
    m = fmaxf(m, fabsf(v));
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(amax, __float_as_uint(red[0]));
}

// amp_obs [M, in] (any row stride) -> the fp16 planes of s_x * clamp((x - mean) / den) in block layout [M, kh / 32, 2, 32],
// zero padded.  One thread per 4 columns (one 8-B store per plane).  Also snapshots the task reward like disc_scale_pad_kernel.
__global__ __launch_bounds__(kBlock) void disc_scale_split_kernel(const float* __restrict__ x, int64_t row_stride, int64_t M,
                                                                  int k, int kh, const float* __restrict__ mean,
                                                                  const float* __restrict__ den, float clip,
                                                                  const DiscRange* __restrict__ range,
                                                                  const float* __restrict__ amax, uint32_t* __restrict__ blocks,
                                                                  const float* __restrict__ task, float* __restrict__ task_copy) {
  const int q_per_row = kh >> 2;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (task && e < M) task_copy[e] = task[e];
  if (e >= M * q_per_row) return;
  const float s_x = plane_scale(amax ? amax[0] : range->clip);
  const int64_t m = e / q_per_row;
  const int c0 = (int)(e - m * q_per_row) * 4;
  const float* row = x + m * row_stride;
  fv4 sv;
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
    sv[i] = v * s_x;
  }
  // block layout: row m, k-block c0 / 32 = [p0 x 32 | p1 x 32] halves; the thread's four columns are 8 B of each plane
  h4 p0, p1;
  split_planes4(sv, p0, p1);
  _Float16* blk = reinterpret_cast<_Float16*>(blocks) + m * (2 * (int64_t)kh) + (c0 >> 5) * 64 + (c0 & 31);
  *reinterpret_cast<h4*>(blk) = p0;
  *reinterpret_cast<h4*>(blk + 32) = p1;
}

}  // namespace amp

namespace amp {
// layers 1-2 + finalize on a scaled, padded input Xs [rows, k1p]
static int disc_forward(const AmpDisc* h, const float* Xs, int64_t rows, float* H1, float* partial, float scale,
                        const float* task, float task_w, float style_w, float* logits, float* style, float* combined,
                        hipStream_t st) {
  // Tile choice, measured on MI355X with tools/gemm_bench.hip (interleaved rounds, profiles/r01_gemm_variants.txt):
  // 128 x 128 x 16 with one LDS stage at 4 workgroups per CU wins whenever it yields >= 512 workgroups; smaller
  // shards use 64 x 64 tiles so that every CU still gets several workgroups (a 4096-row layer 2 is only 128 tiles
  // of 128 x 128: 74 us vs 40 us).
  auto big_tiles = [&](int N) { return (rows + 127) / 128 * (N / 128) >= 512; };
  int rc = AMP_OK;
  GemmArgs g1{};
  g1.A = Xs; g1.lda = h->k1p; g1.M = rows; g1.K = h->k1p;
  g1.W = h->w1p; g1.Kp = h->k1p; g1.bias = h->b1; g1.N = h->h1;
  g1.C = H1; g1.ldc = h->h1;
  {
    const bool big = big_tiles(h->h1);
    const int bm = big ? 128 : 64;
    g1.n_tiles = h->h1 / bm; g1.m_tiles = (int)((rows + bm - 1) / bm);
    const unsigned grid = (unsigned)(((int64_t)g1.m_tiles * g1.n_tiles + 7) / 8 * 8);
    amp::TraceScope trace__("disc_gemm_kernel<0>", st);
    if (big) disc_gemm_kernel<128, 128, 16, 1, 0, 4><<<grid, kBlock, 0, st>>>(g1);
    else disc_gemm_kernel<64, 64, 16, 1, 0, 8><<<grid, kBlock, 0, st>>>(g1);
  }
  rc = launch_status("disc_gemm_kernel<0>");
  if (rc != AMP_OK) return rc;

  GemmArgs g2{};
  g2.A = H1; g2.lda = h->h1; g2.M = rows; g2.K = h->h1;
  g2.W = h->w2; g2.Kp = h->h1; g2.bias = h->b2; g2.N = h->h2;
  g2.w3 = h->w3; g2.partial = partial;
  {
    const bool big = big_tiles(h->h2);
    const int bm = big ? 128 : 64;
    g2.n_tiles = h->h2 / bm; g2.m_tiles = (int)((rows + bm - 1) / bm);
    const unsigned grid = (unsigned)(((int64_t)g2.m_tiles * g2.n_tiles + 7) / 8 * 8);
    amp::TraceScope trace__("disc_gemm_kernel<1>", st);
    if (big) disc_gemm_kernel<128, 128, 16, 1, 1, 4><<<grid, kBlock, 0, st>>>(g2);
    else disc_gemm_kernel<64, 64, 32, 1, 1, 4><<<grid, kBlock, 0, st>>>(g2);
  }
  rc = launch_status("disc_gemm_kernel<1>");
  if (rc != AMP_OK) return rc;

  { amp::TraceScope trace__("disc_finalize_kernel", st);
    disc_finalize_kernel<<<(unsigned)((rows + kBlock - 1) / kBlock), kBlock, 0, st>>>(partial, g2.n_tiles, h->b3, rows, scale, task,
                                                                                   task_w, style_w, logits, style, combined);
  }
  return launch_status("disc_finalize_kernel");
}
// fp16-split forward on the block-layout planes Xp [rows][k1h / 32][2][32] of the scaled input; `amax` = the dynamic
// bound of the scaled input, or null when the scaler's clamp bounds it
template <int TM, int TN, int BK, int MODE, int MINW>
static int launch_f16(GemmF16Args g, int64_t rows, int N, const char* name, hipStream_t st) {
  g.n_tiles = N / (64 * TN);
  g.m_tiles = (int)((rows + 64 * TM - 1) / (64 * TM));
  const unsigned grid = (unsigned)(((int64_t)g.m_tiles * g.n_tiles + 7) / 8 * 8);
  amp::TraceScope trace__(name, st);
  disc_gemm_f16_kernel<TM, TN, BK, MODE, MINW><<<grid, kBlock, gemm_f16_lds_bytes<TM, TN, BK>(), st>>>(g);
  return launch_status(name);
}
// CUs of the current device, rounded down to a multiple of 8 (the tile order of a persistent grid keeps a workgroup's tiles on
// its XCD only when the grid is a multiple of 8); cached per device
static int dma_cu_count() {
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    cus[dev] = n / 8 * 8;
  }
  return cus[dev];
}
// LDS-DMA launch of one layer on a (64 TM) x (128 TN) tile (block-layout operands)
template <int MODE, int TM, int TN, int KB2 = 0>
static int launch_dma(GemmF16Args g, int64_t rows, int N, const char* name, hipStream_t st) {
  using T = DmaTile<TM, TN, KB2 ? KB2 : 2>;
  g.n_tiles = N / T::BN;
  g.m_tiles = (int)((rows + T::BM - 1) / T::BM);
  unsigned grid = (unsigned)(((int64_t)g.m_tiles * g.n_tiles + 7) / 8 * 8);
  // layer 1 (MODE 0): persistent workgroups -- one per LDS slot of the chip, each walking its tiles vb, vb + grid, ... with the
  // next tile's first k-blocks in flight under the current tile's epilogue (no 3-us workgroup hand-over, no exposed first fill)
  if (MODE == 0) grid = std::min(grid, (unsigned)(dma_cu_count() * T::kWgPerCu));
  amp::TraceScope trace__(name, st);
  disc_gemm_f16_dma_kernel<MODE, TM, TN, KB2><<<grid, kDmaThreads, T::kLds, st>>>(g);
  return launch_status(name);
}
template <int MODE, int TM, int TN, int KB2 = 0>
static hipError_t dma_kernel_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gemm_f16_dma_kernel<MODE, TM, TN, KB2>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, DmaTile<TM, TN, KB2 ? KB2 : 2>::kLds);
}
// kernels whose LDS tile exceeds the 64 KB default need the limit raised once per device (not capturable: done at create)
static int f16_kernels_init() {
  // the attribute is per DEVICE: a process that creates discriminators on several GPUs raises the limit on each
  static bool done[64] = {};
  int dev = 0;
  AMP_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && done[dev]) return AMP_OK;
  AMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gemm_f16_kernel<2, 2, 64, 1, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, gemm_f16_lds_bytes<2, 2, 64>()));
  AMP_HIP((dma_kernel_init<0, 4, 2>()));
  AMP_HIP((dma_kernel_init<1, 4, 2>()));
  AMP_HIP((dma_kernel_init<1, 4, 1>()));
  AMP_HIP((dma_kernel_init<0, 2, 1>()));
  AMP_HIP((dma_kernel_init<1, 2, 1>()));
  AMP_HIP((dma_kernel_init<1, 2, 1, 4>()));
  AMP_HIP((dma_kernel_init<1, 1, 1, 4>()));
  AMP_HIP((dma_kernel_init<1, 1, 1>()));
  AMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_mlp_fused_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              FusedLds<4>::kBytes));
  AMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_mlp_fused_kernel<5, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              FusedLds<5>::kBytes));
  AMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_mlp_fused_kernel<6, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              FusedLds<6>::kBytes));
  AMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_mlp_fused_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              FusedLds<4>::kBytes));
  AMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_mlp_fused_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              FusedLds<5>::kBytes));
  AMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(disc_mlp_fused_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              FusedLds<6>::kBytes));
  if (dev >= 0 && dev < 64) done[dev] = true;
  return AMP_OK;
}
// Row chunking of a large shard: layer 1 writes the hidden layer of a chunk (4 KB / row) and layer 2 reads it back
// right away, so a chunk that fits the 256 MB Infinity Cache next to everything else touched in between is served
// on-die instead of from HBM (a 65 536-row shard writes 268 MB: read back in the same order NOTHING would still be
// resident).  32 768 rows = 134 MB, and 128 x 2 layer-2 tiles = one workgroup per CU.
constexpr int64_t kChunkRows = 32768;

// Kernel plan of a chunk of `rows` rows, measured on MI355X with tools/gemm_f16_bench.hip (TILES=1 / default mode;
// profiles/r02_gemm_f16_small_shards.txt).  What decides it: an LDS-DMA workgroup takes in ~68 GB/s, so a tile must be
// large enough to compute longer than its fills take (256 x 256: 21 B of fill per matrix-pipe cycle of the CU, 128 x 128: 42)
// AND the launch must cover the 256 CUs.  us per launch, layer 1 / layer 2 (K = 192 / 1024):
//   rows      256x256      256x128      128x128      register-staged 128x128 / 64x64
//   32 768    56 / 80
//   16 384    28.9 / 66.7   33.6 / 44.3  30.8 / 47.1
//   12 288    26.6 / 64.2   31.1 / 40.6  25.4 / 42.8
//    8 192    24.8 / 62.0   19.1 / 37.9  18.0 / 29.0   18.5 / 34.6, 20.8 / 37.6
//    4 096    22.7 / 60.2   16.1 / 35.6  12.0 / 25.9   (64 x 64: 12 / 22 inside the step; 64 x 128 LDS-DMA layer 2: 20.4, ~18 on a 4-stage ring)
enum { kPlanRegister = 0, kPlanDmaSmall = 1, kPlanDmaMid = 2, kPlanDmaLarge = 3, kPlanDmaTiny = 4, kPlanFused = 5 };
// Fused two-layer kernel (disc_mlp_fused.hpp): one workgroup per 128 rows and all 512 output columns, so it needs at least one
// workgroup per CU to pay (AMP_DISC_FUSED_MIN_ROWS overrides the threshold, AMP_DISC_FUSED=0 switches the plan off: A/B runs and
// the plan-independence tests); shapes: h2 = 512 (the accumulator tile), K D padded to 192 (activation fragments in registers).
static int64_t fused_min_rows(const AmpDisc* h) {
  if (h && h->ovr_fused_min_rows >= 0) return h->ovr_fused_min_rows;
  static int64_t v = -1;
  if (v < 0) {
    const char* off = getenv("AMP_DISC_FUSED");
    const char* mn = getenv("AMP_DISC_FUSED_MIN_ROWS");
    v = (off && off[0] == '0') ? INT64_MAX : (mn && atoll(mn) >= kFusedRows ? atoll(mn) : 24576);
  }
  return v;
}
// Leading rows of a `rows`-row batch that take the fused kernel.  A 128-row tile runs ~100 us whatever the batch, so the launch
// costs whole ROUNDS of workgroups: every full round (one tile per CU: 32 768 rows on 256 CUs) is fused, and the rest joins it
// only if it fills a last round to at least the threshold (24 576 rows: three quarters) -- otherwise those rows go through the
// column-split two-kernel plans, which scale down (measured, us per step fused / split / unfused: 40 960 rows 231.9 / - / 225.3,
// 49 152 rows 243.6 / - / 241.6 before the split; profiles/r04_fused_mlp_kernel.md).  Any split gives the same bits per row.
static int64_t fused_rows_of(const AmpDisc* h, int64_t rows) {
  // input widths K D in (96, 192]: 4, 5 or 6 k-blocks of activation fragments resident in registers (KX = k1h / 32)
  if (!(h->h2 == kFusedN2 && h->k1h >= 128 && h->k1h <= 192 && h->h1 % 32 == 0 && h->h1 <= 1024) || rows < fused_min_rows(h) || rows < kFusedRows) return 0;
  const int64_t per_round = (int64_t)dma_cu_count() * kFusedRows;
  const int64_t full = rows / per_round * per_round, rem = rows - full;
  return rem >= fused_min_rows(h) || rem == 0 ? rows : full;
}
static int f16_plan(const AmpDisc* h, int64_t rows) {
  if (h->h1 % kDmaBN != 0 || h->h2 % kDmaBN != 0) return kPlanRegister;   // the LDS-DMA tiles need 256-column multiples
  if ((rows + kDmaBM - 1) / kDmaBM * (h->h2 / kDmaBN) >= 192) return kPlanDmaLarge;  // >= ~1 tile of 256 x 256 per CU
  // layer 1: 256 x 256, layer 2: 256 x 128 -- only where those tiles fill the chip in whole rounds (16 384 rows: 256 layer-2
  // tiles); between 16 384 and 24 576 rows the 128 x 128 tiles' finer granularity wins (20 000 envs: 142 -> 124 us per step,
  // same box, round 3)
  if (rows >= 12288 && rows <= 16384) return kPlanDmaMid;
  if (rows > 5120) return kPlanDmaSmall;     // both layers 128 x 128 (the 8 192-env shards of the multi-GPU configurations)
  // layer 1 128 x 128, layer 2 64 x 128 on the four-stage k-block ring (one workgroup per CU at 4 096 rows): the 4 096-env
  // configuration, 47.2 -> 43.6 us per step against the register-staged 64 x 64 tiles (same box, tools/ab_bench.sh, round 3)
  if (rows >= 3072) return kPlanDmaTiny;
  return kPlanRegister;
}
static bool f16_use_dma(const AmpDisc* h, int64_t rows) { return f16_plan(h, rows) != kPlanRegister; }
static int64_t f16_chunk_rows(const AmpDisc* h, int64_t rows) {
  return f16_plan(h, kChunkRows) == kPlanDmaLarge && rows > kChunkRows + kChunkRows / 2 ? kChunkRows : rows;
}
// fp32 observation rows handed straight to the fused two-layer kernel (its RAWX instantiation scales / clamps / splits them itself)
struct RawRows { const float* x; int64_t ld; };
// ... possible when the whole batch takes the fused kernel, a clamping scaler bounds the input, and the rows are 8-B aligned pairs
static bool raw_rows_ok(const AmpDisc* h, int64_t rows, const float* x, int64_t row_stride);

static int disc_forward_f16(const AmpDisc* h, const _Float16* Xp, const float* amax, int64_t rows, _Float16* H1p, float* partial,
                            float scale, const float* task, float task_w, float style_w, float* logits, float* style,
                            float* combined, hipStream_t st, const CompactLaunch* compact = nullptr, const RawRows* raw = nullptr) {
  // every kernel below accumulates in the same order and emits the same canonical partial logits (one per row and
  // 32-column block), so the choice changes the time, never a bit of the result
  const int64_t n_fused = fused_rows_of(h, rows);  // leading rows on the fused two-layer kernel (no hidden layer in memory)
  const int64_t chunk = f16_chunk_rows(h, rows - n_fused);
  const int n_blocks = h->h2 / 32;  // canonical partial logits: one per (row, 32-column block)
  int rc = AMP_OK;
  if (n_fused > 0) {
    FusedArgs f{};
    f.X = Xp; f.ldx = h->k1h; f.M = n_fused;
    f.W1b = h->w1b; f.W2b = h->w2b; f.b1 = h->b1; f.b2 = h->b2; f.w3 = h->w3;
    f.range = h->range; f.amax = amax; f.h1 = h->h1; f.partial = partial;
    const unsigned grid = (unsigned)((n_fused + kFusedRows - 1) / kFusedRows);
    if (raw) {
      AMP_REQUIRE(n_fused == rows, "disc_forward_f16: raw rows need the whole batch on the fused kernel");
      f.raw = raw->x; f.raw_ld = raw->ld; f.raw_cols = h->in_dim; f.mean = h->mean; f.den = h->den; f.clip = h->clip;
      f.s_x = plane_scale(h->clip);
      amp::TraceScope trace__("disc_mlp_fused_kernel", st);
      if (h->k1h == 128) disc_mlp_fused_kernel<4, true><<<grid, kFusedThreads, FusedLds<4>::kBytes, st>>>(f);
      else if (h->k1h == 160) disc_mlp_fused_kernel<5, true><<<grid, kFusedThreads, FusedLds<5>::kBytes, st>>>(f);
      else disc_mlp_fused_kernel<6, true><<<grid, kFusedThreads, FusedLds<6>::kBytes, st>>>(f);
    } else {
      amp::TraceScope trace__("disc_mlp_fused_kernel", st);
      if (h->k1h == 128) disc_mlp_fused_kernel<4><<<grid, kFusedThreads, FusedLds<4>::kBytes, st>>>(f);
      else if (h->k1h == 160) disc_mlp_fused_kernel<5><<<grid, kFusedThreads, FusedLds<5>::kBytes, st>>>(f);
      else disc_mlp_fused_kernel<6><<<grid, kFusedThreads, FusedLds<6>::kBytes, st>>>(f);
    }
    rc = launch_status("disc_mlp_fused_kernel");
    if (rc != AMP_OK) return rc;
  }
  for (int64_t r0 = n_fused; r0 < rows && rc == AMP_OK; r0 += chunk) {
    const int64_t m = rows - r0 < chunk ? rows - r0 : chunk;
    const int plan = f16_plan(h, chunk);  // by the chunk size (a short last chunk keeps the hidden layer's layout)
    const bool dma = plan != kPlanRegister;
    auto big_tiles = [&](int N) { return (m + 127) / 128 * (N / 128) >= 512; };
    GemmF16Args g1{};
    g1.A = Xp + 2 * r0 * h->k1h; g1.lda = h->k1h; g1.plane_a = 0; g1.M = m;  // block layout: 2 * k1h halves per row
    g1.W = h->w1h; g1.plane_w = (int64_t)h->h1 * h->k1h; g1.Kp = h->k1h; g1.N = h->h1;
    g1.bias = h->b1; g1.range = h->range; g1.amax = amax; g1.layer = 1;
    g1.H = H1p + r0 * h->h1; g1.ldh = h->h1; g1.plane_h = rows * h->h1;
    g1.ksteps = (h->in_dim + 15) / 16;  // k-steps beyond the true K hold zero padding in both operands: skipped
    if (dma) {
      g1.W = h->w1b;  // block layout; the hidden layer comes out in block layout too (row pitch 2 * h1)
      g1.H = H1p;     // every chunk reuses the SAME 134 MB: rewritten while still dirty in the Infinity Cache, the hidden
                      // layer is (mostly) never written back to HBM, and the lines it evicts are not dirty either
      if (plan == kPlanDmaSmall || plan == kPlanDmaTiny) rc = launch_dma<0, 2, 1>(g1, m, h->h1, "disc_gemm_f16_dma_kernel<0>", st);
      else rc = launch_dma<0, 4, 2>(g1, m, h->h1, "disc_gemm_f16_dma_kernel<0>", st);
    } else if (big_tiles(h->h1)) {
      rc = launch_f16<2, 2, 32, 0, 3>(g1, m, h->h1, "disc_gemm_f16_kernel<0>", st);
    } else {
      rc = launch_f16<1, 1, 32, 0, 6>(g1, m, h->h1, "disc_gemm_f16_kernel<0>", st);
    }
    if (rc != AMP_OK) return rc;

    GemmF16Args g2{};
    g2.A = H1p + r0 * h->h1; g2.lda = h->h1; g2.plane_a = rows * h->h1; g2.M = m;
    g2.W = h->w2h; g2.plane_w = (int64_t)h->h2 * h->h1; g2.Kp = h->h1; g2.N = h->h2;
    g2.bias = h->b2; g2.range = h->range; g2.amax = amax; g2.layer = 2;
    g2.w3 = h->w3; g2.partial = partial + r0 * n_blocks;
    if (dma) {
      g2.A = H1p;
      g2.W = h->w2b;
      // 128 x 128 tiles: when the launch is ONE round of workgroups (<= one per CU: the 8 192-env shards) the k-block-per-
      // segment schedule on a four-stage ring (128 KB: one workgroup per CU anyway) is ~3 us faster (8 192 rows 31.5 -> 29.1 us
      // with three stages, 28.2 with four; 6 144 rows 29.9 -> 26.8, engine tracer) -- what helps is the extra fill cover: the same
      // schedule on TWO stages measured 31.8 us; with more tiles than CUs two 64-KB workgroups per CU win (12 000 rows: 45.9 vs
      // 49.3 us).  Same accumulation order: bit-identical either way.
      const bool one_round = (m + 127) / 128 * (h->h2 / 128) <= dma_cu_count();
      if (plan == kPlanDmaSmall) rc = one_round ? launch_dma<1, 2, 1, 4>(g2, m, h->h2, "disc_gemm_f16_dma_kernel<1>", st)
                                                : launch_dma<1, 2, 1>(g2, m, h->h2, "disc_gemm_f16_dma_kernel<1>", st);
      else if (plan == kPlanDmaTiny) rc = (m + 63) / 64 * (h->h2 / 128) <= dma_cu_count() ? launch_dma<1, 1, 1, 4>(g2, m, h->h2, "disc_gemm_f16_dma_kernel<1>", st)
                                                                              : launch_dma<1, 1, 1>(g2, m, h->h2, "disc_gemm_f16_dma_kernel<1>", st);
      else if (plan == kPlanDmaMid) rc = launch_dma<1, 4, 1>(g2, m, h->h2, "disc_gemm_f16_dma_kernel<1>", st);
      else rc = launch_dma<1, 4, 2>(g2, m, h->h2, "disc_gemm_f16_dma_kernel<1>", st);
    } else if (big_tiles(h->h2)) {
      rc = launch_f16<2, 2, 64, 1, 2>(g2, m, h->h2, "disc_gemm_f16_kernel<1>", st);
    } else {
      rc = launch_f16<1, 1, 64, 1, 4>(g2, m, h->h2, "disc_gemm_f16_kernel<1>", st);
    }
  }
  if (rc != AMP_OK) return rc;
  const unsigned fin_blocks = (unsigned)((rows + kBlock - 1) / kBlock);
  if (compact) {
    amp::TraceScope trace__("step_tail_kernel", st);
    step_tail_kernel<<<fin_blocks + (unsigned)compact->blocks, kBlock, 0, st>>>(*compact, partial, n_blocks, h->b3, rows, scale, task,
                                                                              task_w, style_w, logits, style, combined);
    return launch_status("step_tail_kernel");
  }
  { amp::TraceScope trace__("disc_finalize_kernel", st);
    disc_finalize_kernel<<<fin_blocks, kBlock, 0, st>>>(partial, n_blocks, h->b3, rows, scale, task, task_w, style_w, logits, style,
                                                      combined);
  }
  return launch_status("disc_finalize_kernel");
}

// weights (or the scaler) changed: ranges, plane scales and the fp16 planes of W1 / W2
// W1 [h1, k1p] -> planes [h1, k1h] (plain: w1h + plane n1; block layout: w1b) and W2 [h2, h1] likewise, one thread per four
// columns, one read of the weights for both layouts (the bodies of split_rows_f16_kernel and split_rows_blocks_kernel)
__global__ __launch_bounds__(kBlock) void split_weights_kernel(const float* __restrict__ w1, int h1, int k1p, int k1h,
                                                               _Float16* __restrict__ w1h, _Float16* __restrict__ w1b, int64_t n1,
                                                               const float* __restrict__ w2, int h2, int k2,
                                                               _Float16* __restrict__ w2h, _Float16* __restrict__ w2b, int64_t n2,
                                                               const DiscRange* __restrict__ range, unsigned blocks1) {
  const bool first = blockIdx.x < blocks1;
  const float* src = first ? w1 : w2;
  const int rows = first ? h1 : h2, cols = first ? k1p : k2, kp = first ? k1h : k2;   // cols = source row pitch = valid columns
  _Float16* plain = first ? w1h : w2h;
  _Float16* blk_out = first ? w1b : w2b;
  const int64_t plane = first ? n1 : n2;
  const float s = first ? range->s_w1 : range->s_w2;
  const int64_t q = (int64_t)(first ? blockIdx.x : blockIdx.x - blocks1) * kBlock + threadIdx.x, per_row = kp / 4;
  if (q >= (int64_t)rows * per_row) return;
  const int64_t r = q / per_row;
  const int c = (int)(q - r * per_row) * 4;
  fv4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = c + i < cols ? src[r * cols + c + i] * s : 0.0f;
  h4 p0, p1;
  split_planes4(v, p0, p1);
  *reinterpret_cast<h4*>(&plain[r * kp + c]) = p0;
  *reinterpret_cast<h4*>(&plain[plane + r * kp + c]) = p1;
  _Float16* blk = blk_out + r * (2 * (int64_t)kp) + (c >> 5) * 64 + (c & 31);
  *reinterpret_cast<h4*>(blk) = p0;
  *reinterpret_cast<h4*>(blk + 32) = p1;
}

// disc_weight_range_kernel keeps its accumulators and workgroup ticket in the record between launches (the finisher clears them).
// A launch that faulted, or two refreshes of one handle in flight on different streams, would leave them non-zero and every later
// refresh would publish partial maxima: the entries off the hot path (amp_disc_set_weights, amp_disc_trainer_create) re-zero them.
int disc_range_reset(AmpDisc* h, hipStream_t st) {
  AMP_HIP(hipMemsetAsync(reinterpret_cast<char*>(h->range) + offsetof(DiscRange, raw), 0, sizeof(unsigned) * 5, st));
  return AMP_OK;
}

static int f16_refresh(AmpDisc* h, hipStream_t st) {
  { amp::TraceScope trace__("disc_weight_range_kernel", st);
    disc_weight_range_kernel<<<128, kBlock, 0, st>>>(h->w1p, h->h1, h->k1p, h->b1, h->w2, (int64_t)h->h2 * h->h1, h->range);
  }
  int rc = launch_status("disc_weight_range_kernel");
  if (rc != AMP_OK) return rc;
  const int64_t n1 = (int64_t)h->h1 * h->k1h, n2 = (int64_t)h->h2 * h->h1;
  // both weights into both plane layouts in ONE launch (the training step refreshes them after every update: four 5-us launches
  // in a row were a third of that refresh); same values as split_rows_f16_kernel / split_rows_blocks_kernel
  const unsigned b1n = (unsigned)((n1 / 4 + kBlock - 1) / kBlock), b2n = (unsigned)((n2 / 4 + kBlock - 1) / kBlock);
  { amp::TraceScope trace__("split_weights_kernel", st);
    split_weights_kernel<<<b1n + b2n, kBlock, 0, st>>>(h->w1p, h->h1, h->k1p, h->k1h, h->w1h, h->w1b, n1, h->w2, h->h2, h->h1, h->w2h, h->w2b,
                                                     n2, h->range, b1n);
  }
  return launch_status("split_weights_kernel");
}
}  // namespace amp

namespace amp {
// accessors for disc_train.hip
struct DiscParams {
  int32_t in_dim, h1, h2, k1p;
  float *w1p, *b1, *w2, *b2, *w3, *b3;
};
DiscParams disc_params(AmpDisc* h) { return DiscParams{h->in_dim, h->h1, h->h2, h->k1p, h->w1p, h->b1, h->w2, h->b2, h->w3, h->b3}; }
int disc_refresh_derived(AmpDisc* h, hipStream_t st) {
  // the fp32 weights changed in place: ranges and fp16 planes follow
  return f16_refresh(h, st);
}
}  // namespace amp

using namespace amp;

extern "C" {

int amp_disc_destroy(AmpDisc* h) {
  if (!h) return AMP_OK;
  (void)hipFree(h->w1p);
  (void)hipFree(h->b1);
  (void)hipFree(h->w2);
  (void)hipFree(h->b2);
  (void)hipFree(h->w3);
  (void)hipFree(h->b3);
  (void)hipFree(h->mean);
  (void)hipFree(h->den);
  (void)hipFree(h->w1h);
  (void)hipFree(h->w2h);
  (void)hipFree(h->w1b);
  (void)hipFree(h->w2b);
  (void)hipFree(h->range);
  delete h;
  return AMP_OK;
}

int amp_disc_create(const AmpDiscDesc* d, amp_stream_t stream, AmpDisc** out) {
  AMP_REQUIRE(d && out, "amp_disc_create: null argument");
  AMP_REQUIRE(d->in_dim >= 1, "amp_disc_create: in_dim must be positive");
  AMP_REQUIRE(d->h1 >= kMinN && d->h1 % kMinN == 0 && d->h1 % kPadK == 0, "amp_disc_create: h1=%d must be a multiple of %d", d->h1, kMinN);
  AMP_REQUIRE(d->h2 >= kMinN && d->h2 % kMinN == 0, "amp_disc_create: h2=%d must be a multiple of %d", d->h2, kMinN);
  AMP_REQUIRE(d->w1 && d->b1 && d->w2 && d->b2 && d->w3 && d->b3, "amp_disc_create: null weight pointer");
  AmpDisc* h = new (std::nothrow) AmpDisc();
  AMP_REQUIRE(h, "amp_disc_create: out of host memory");
  *h = AmpDisc{};
  h->in_dim = d->in_dim;
  h->h1 = d->h1;
  h->h2 = d->h2;
  h->k1p = (int32_t)round_up(d->in_dim, kPadK);
  h->k1h = (int32_t)round_up(d->in_dim, kPadKH);
  h->mode = AMP_DISC_F16X3;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMalloc(&h->w1p, sizeof(float) * (size_t)h->h1 * h->k1p);
  if (e == hipSuccess) e = hipMalloc(&h->b1, sizeof(float) * h->h1);
  if (e == hipSuccess) e = hipMalloc(&h->w2, sizeof(float) * (size_t)h->h2 * h->h1);
  if (e == hipSuccess) e = hipMalloc(&h->b2, sizeof(float) * h->h2);
  if (e == hipSuccess) e = hipMalloc(&h->w3, sizeof(float) * h->h2);
  if (e == hipSuccess) e = hipMalloc(&h->b3, sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&h->mean, sizeof(float) * h->k1p);
  if (e == hipSuccess) e = hipMalloc(&h->den, sizeof(float) * h->k1p);
  if (e == hipSuccess) e = hipMalloc(&h->w1h, sizeof(_Float16) * 2 * (size_t)h->h1 * h->k1h);
  if (e == hipSuccess) e = hipMalloc(&h->w2h, sizeof(_Float16) * 2 * (size_t)h->h2 * h->h1);
  if (e == hipSuccess) e = hipMalloc(&h->w1b, sizeof(_Float16) * 2 * (size_t)h->h1 * h->k1h);
  if (e == hipSuccess) e = hipMalloc(&h->w2b, sizeof(_Float16) * 2 * (size_t)h->h2 * h->h1);
  if (e == hipSuccess) e = hipMalloc(&h->range, sizeof(DiscRange));
  if (e == hipSuccess) e = hipMemsetAsync(h->range, 0, sizeof(DiscRange), st);
  if (e == hipSuccess) e = hipMemcpyAsync(h->b1, d->b1, sizeof(float) * h->h1, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(h->w2, d->w2, sizeof(float) * (size_t)h->h2 * h->h1, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(h->b2, d->b2, sizeof(float) * h->h2, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(h->w3, d->w3, sizeof(float) * h->h2, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(h->b3, d->b3, sizeof(float), hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) {
    amp_disc_destroy(h);
    return fail(AMP_ERR_HIP, "amp_disc_create: %s", hipGetErrorString(e));
  }
  const int64_t total = (int64_t)h->h1 * h->k1p;
  { amp::TraceScope trace__("disc_pad_rows_kernel", st);
    disc_pad_rows_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(d->w1, h->h1, h->in_dim, h->k1p, h->w1p);
  }
  int rc = launch_status("disc_pad_rows_kernel");
  if (rc == AMP_OK) rc = f16_kernels_init();
  if (rc == AMP_OK) rc = f16_refresh(h, st);
  if (rc == AMP_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(AMP_ERR_HIP, "amp_disc_create: stream sync failed");
  if (rc != AMP_OK) {
    amp_disc_destroy(h);
    return rc;
  }
  *out = h;
  return AMP_OK;
}

int amp_disc_set_weights(AmpDisc* h, const AmpDiscDesc* d, amp_stream_t stream) {
  AMP_REQUIRE(h && d, "amp_disc_set_weights: null argument");
  AMP_REQUIRE(d->in_dim == h->in_dim && d->h1 == h->h1 && d->h2 == h->h2,
              "amp_disc_set_weights: shape [%d,%d,%d] does not match the handle's [%d,%d,%d]", d->in_dim, d->h1, d->h2, h->in_dim,
              h->h1, h->h2);
  AMP_REQUIRE(d->w1 && d->b1 && d->w2 && d->b2 && d->w3 && d->b3, "amp_disc_set_weights: null weight pointer");
  hipStream_t st = (hipStream_t)stream;
  // device-to-device copies into the handle's own buffers: no allocation, no sync; scaler, precision and every device
  // pointer handed out earlier (amp_disc_input_layout, a trainer's view of the weights) stay valid
  AMP_HIP(hipMemcpyAsync(h->b1, d->b1, sizeof(float) * h->h1, hipMemcpyDeviceToDevice, st));
  AMP_HIP(hipMemcpyAsync(h->w2, d->w2, sizeof(float) * (size_t)h->h2 * h->h1, hipMemcpyDeviceToDevice, st));
  AMP_HIP(hipMemcpyAsync(h->b2, d->b2, sizeof(float) * h->h2, hipMemcpyDeviceToDevice, st));
  AMP_HIP(hipMemcpyAsync(h->w3, d->w3, sizeof(float) * h->h2, hipMemcpyDeviceToDevice, st));
  AMP_HIP(hipMemcpyAsync(h->b3, d->b3, sizeof(float), hipMemcpyDeviceToDevice, st));
  const int64_t total = (int64_t)h->h1 * h->k1p;
  { amp::TraceScope trace__("disc_pad_rows_kernel", st);
    disc_pad_rows_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(d->w1, h->h1, h->in_dim, h->k1p, h->w1p);
  }
  int rc = launch_status("disc_pad_rows_kernel");
  if (rc != AMP_OK) return rc;
  rc = disc_range_reset(h, st);
  if (rc != AMP_OK) return rc;
  return f16_refresh(h, st);
}

int amp_disc_set_scaler(AmpDisc* h, const double* mean, const double* var, float eps, float clip, amp_stream_t stream) {
  AMP_REQUIRE(h, "amp_disc_set_scaler: null handle");
  if (!mean) {
    h->has_scaler = false;
    return AMP_OK;
  }
  AMP_REQUIRE(var, "amp_disc_set_scaler: running_variance is null");
  { amp::TraceScope trace__("disc_scaler_kernel", (hipStream_t)stream);
    disc_scaler_kernel<<<(h->k1p + 255) / 256, 256, 0, (hipStream_t)stream>>>(mean, var, h->in_dim, h->k1p, eps, h->mean, h->den);
  }
  int rc = launch_status("disc_scaler_kernel");
  if (rc != AMP_OK) return rc;
  disc_set_clip_kernel<<<1, 1, 0, (hipStream_t)stream>>>(h->range, clip);
  rc = launch_status("disc_set_clip_kernel");
  if (rc != AMP_OK) return rc;
  h->clip = clip;
  h->has_scaler = true;
  return AMP_OK;
}

int amp_disc_set_precision(AmpDisc* h, int32_t mode, amp_stream_t stream) {
  (void)stream;
  AMP_REQUIRE(h, "amp_disc_set_precision: null handle");
  AMP_REQUIRE(mode == AMP_DISC_F16X3 || mode == AMP_DISC_FP32, "amp_disc_set_precision: mode must be AMP_DISC_F16X3 (0) or AMP_DISC_FP32 (1)");
  h->mode = mode;
  return AMP_OK;
}

int amp_disc_get_weights(const AmpDisc* h, float* w1, float* b1, float* w2, float* b2, float* w3, float* b3, amp_stream_t stream) {
  AMP_REQUIRE(h && w1 && b1 && w2 && b2 && w3 && b3, "amp_disc_get_weights: null argument");
  hipStream_t st = (hipStream_t)stream;
  // W1 is stored zero-padded to k1p columns: copy the logical [h1, in_dim] block
  AMP_HIP(hipMemcpy2DAsync(w1, sizeof(float) * h->in_dim, h->w1p, sizeof(float) * h->k1p, sizeof(float) * h->in_dim, h->h1,
                           hipMemcpyDeviceToDevice, st));
  AMP_HIP(hipMemcpyAsync(b1, h->b1, sizeof(float) * h->h1, hipMemcpyDeviceToDevice, st));
  AMP_HIP(hipMemcpyAsync(w2, h->w2, sizeof(float) * (size_t)h->h2 * h->h1, hipMemcpyDeviceToDevice, st));
  AMP_HIP(hipMemcpyAsync(b2, h->b2, sizeof(float) * h->h2, hipMemcpyDeviceToDevice, st));
  AMP_HIP(hipMemcpyAsync(w3, h->w3, sizeof(float) * h->h2, hipMemcpyDeviceToDevice, st));
  AMP_HIP(hipMemcpyAsync(b3, h->b3, sizeof(float), hipMemcpyDeviceToDevice, st));
  return AMP_OK;
}

namespace amp {
// workspace: [header][Xs: fp32 [rows, k1p] or planes [2][rows, k1h]][H1: fp32 or planes, 4 B / element][partial][task]
struct DiscWorkspace {
  float* header; void* xs; void* h1; float* partial; float* task_copy; int64_t bytes;
};
static DiscWorkspace disc_workspace(const AmpDisc* h, int64_t rows, void* base) {
  DiscWorkspace w;
  char* p = (char*)base;
  w.header = (float*)p; p += kWsHeader;
  w.xs = p; p += round_up(4 * rows * (h->k1h > h->k1p ? h->k1h : h->k1p), 256);
  w.h1 = p; p += round_up(4 * rows * h->h1, 256);
  w.partial = (float*)p; p += round_up((int64_t)sizeof(float) * rows * (h->h2 / 32), 256);
  w.task_copy = (float*)p; p += round_up((int64_t)sizeof(float) * rows, 256);
  w.bytes = p - (char*)base;
  return w;
}
// the clamp bounds the scaled input only when a scaler with a finite positive clamp is set
static bool static_bound(const AmpDisc* h) { return h->has_scaler && h->clip > 0.0f && h->clip < 1e30f; }
static bool raw_rows_ok(const AmpDisc* h, int64_t rows, const float* x, int64_t row_stride) {
  return h->mode == AMP_DISC_F16X3 && static_bound(h) && rows >= kFusedRows && fused_rows_of(h, rows) == rows && (h->in_dim & 1) == 0 &&
         (row_stride & 1) == 0 && ((uintptr_t)x & 7) == 0;
}
}  // namespace amp

int64_t amp_disc_workspace_bytes(const AmpDisc* h, int64_t rows) {
  if (!h || rows < 0) return -1;
  return disc_workspace(h, rows, nullptr).bytes;
}

int amp_disc_style_reward(const AmpDisc* h, const float* x, int64_t rows, int64_t row_stride, float scale, const float* task,
                          float task_w, float style_w, float* logits, float* style, float* combined, void* workspace,
                          amp_event_t inputs_consumed, amp_stream_t stream) {
  AMP_REQUIRE(h, "amp_disc_style_reward: null handle");
  AMP_REQUIRE(rows >= 0, "amp_disc_style_reward: negative rows");
  if (rows == 0) return AMP_OK;
  AMP_REQUIRE(x && workspace, "amp_disc_style_reward: null buffer");
  AMP_REQUIRE(row_stride >= h->in_dim, "amp_disc_style_reward: row_stride %lld < in_dim %d", (long long)row_stride, h->in_dim);
  AMP_REQUIRE((uintptr_t)workspace % 16 == 0, "amp_disc_style_reward: workspace must be 16-byte aligned");
  AMP_REQUIRE(rows <= ((int64_t)1 << 30), "amp_disc_style_reward: too many rows");
  hipStream_t st = (hipStream_t)stream;
  const DiscWorkspace ws = disc_workspace(h, rows, workspace);
  const float* mean = h->has_scaler ? h->mean : nullptr;
  if (raw_rows_ok(h, rows, x, row_stride)) {
    // the whole batch takes the fused two-layer kernel: it reads the fp32 rows itself (no scaler pass, no scaled copy)
    const RawRows raw{x, row_stride};
    const int rc = disc_forward_f16(h, nullptr, nullptr, rows, (_Float16*)ws.h1, ws.partial, scale, task, task_w, style_w, logits, style,
                                    combined, st, nullptr, &raw);
    if (rc == AMP_OK && inputs_consumed) AMP_HIP(hipEventRecord((hipEvent_t)inputs_consumed, st));  // (here: after the whole call)
    return rc;
  }
  if (h->mode == AMP_DISC_F16X3) {
    const float* amax = nullptr;
    if (!static_bound(h)) {  // no clamp: the bound of the scaled input is its abs-max (one extra pass)
      AMP_HIP(hipMemsetAsync(ws.header, 0, sizeof(float), st));
      amp::TraceScope trace__("disc_absmax_kernel", st);
      disc_absmax_kernel<<<1024, kBlock, 0, st>>>(x, row_stride, rows, h->in_dim, mean, h->den, h->clip, (unsigned*)ws.header);
      amax = ws.header;
    }
    {
      const int64_t quads = rows * (h->k1h / 4);
      amp::TraceScope trace__("disc_scale_split_kernel", st);
      disc_scale_split_kernel<<<(unsigned)((quads + kBlock - 1) / kBlock), kBlock, 0, st>>>(
          x, row_stride, rows, h->in_dim, h->k1h, mean, h->den, h->clip, h->range, amax, (uint32_t*)ws.xs, task, ws.task_copy);
    }
    int rcs = launch_status("disc_scale_split_kernel");
    if (rcs != AMP_OK) return rcs;
    if (inputs_consumed) AMP_HIP(hipEventRecord((hipEvent_t)inputs_consumed, st));
    return disc_forward_f16(h, (const _Float16*)ws.xs, amax, rows, (_Float16*)ws.h1, ws.partial, scale, task ? ws.task_copy : nullptr,
                            task_w, style_w, logits, style, combined, st);
  }
  float* Xs = (float*)ws.xs;
  {
    const int64_t quads = rows * (h->k1p / 4);
    amp::TraceScope trace__("disc_scale_pad_kernel", st);
    disc_scale_pad_kernel<<<(unsigned)((quads + kBlock - 1) / kBlock), kBlock, 0, st>>>(x, row_stride, rows, h->in_dim, h->k1p, mean,
                                                                                      h->den, h->clip, Xs, task, ws.task_copy);
  }
  int rc = launch_status("disc_scale_pad_kernel");
  if (rc != AMP_OK) return rc;
  // everything the caller handed in (amp_obs, task reward) has been consumed once this point of the stream is reached
  if (inputs_consumed) AMP_HIP(hipEventRecord((hipEvent_t)inputs_consumed, st));
  return disc_forward(h, Xs, rows, (float*)ws.h1, ws.partial, scale, task ? ws.task_copy : nullptr, task_w, style_w, logits, style,
                      combined, st);
}

int amp_disc_input_layout(const AmpDisc* h, AmpDiscInputLayout* out) {
  AMP_REQUIRE(h && out, "amp_disc_input_layout: null argument");
  const bool planes = h->mode == AMP_DISC_F16X3 && static_bound(h);
  out->format = planes ? AMP_DISC_INPUT_F16_BLOCKS : AMP_DISC_INPUT_F32_ROWS;
  out->padded_dim = planes ? h->k1h : h->k1p;
  out->mean_dev = h->has_scaler ? h->mean : nullptr;
  out->den_dev = h->den;
  out->clip = h->clip;
  out->plane_scale = planes ? plane_scale(h->clip) : 1.0f;
  return AMP_OK;
}

int amp_disc_set_plan(AmpDisc* h, int32_t fused, int64_t fused_min_rows_) {
  AMP_REQUIRE(h, "amp_disc_set_plan: null handle");
  AMP_REQUIRE(fused >= -1 && fused <= 1, "amp_disc_set_plan: fused must be -1 (automatic), 0 (off) or 1 (on)");
  AMP_REQUIRE(fused_min_rows_ == -1 || fused_min_rows_ >= kFusedRows, "amp_disc_set_plan: fused_min_rows must be -1 or >= %d", kFusedRows);
  h->ovr_fused_min_rows = fused == 0 ? INT64_MAX : (fused_min_rows_ >= 0 ? fused_min_rows_ : (fused == 1 ? 24576 : -1));
  return AMP_OK;
}

int amp_disc_plan_info(const AmpDisc* h, int64_t rows, AmpDiscPlanInfo* out) {
  AMP_REQUIRE(h && out, "amp_disc_plan_info: null argument");
  AMP_REQUIRE(rows >= 0, "amp_disc_plan_info: negative row count");
  *out = AmpDiscPlanInfo{};
  out->precision = h->mode;
  out->cu_count = dma_cu_count();
  out->fused_min_rows = fused_min_rows(h);
  auto set = [](const char* name) { const char* e = getenv(name); return e && e[0]; };
  out->env_overrides = (set("AMP_DISC_FUSED") ? AMP_ENV_DISC_FUSED : 0) | (set("AMP_DISC_FUSED_MIN_ROWS") ? AMP_ENV_DISC_FUSED_MIN_ROWS : 0) |
                       (set("AMP_TRAIN_FORK") ? AMP_ENV_TRAIN_FORK : 0) | (set("AMP_TRAIN_BK32") ? AMP_ENV_TRAIN_BK32 : 0) |
                       (set("AMP_TRAIN_F16_BIG") ? AMP_ENV_TRAIN_F16_BIG : 0);
  if (h->mode != AMP_DISC_F16X3) {
    out->plan = AMP_DISC_PLAN_FP32;
    out->chunk_rows = rows;
    return AMP_OK;
  }
  out->fused_rows = fused_rows_of(h, rows);
  out->raw_input = raw_rows_ok(h, rows, nullptr, 0) ? 1 : 0;
  const int64_t rest = rows - out->fused_rows;
  out->chunk_rows = rest > 0 ? f16_chunk_rows(h, rest) : 0;
  out->plan = f16_plan(h, rest > 0 ? out->chunk_rows : rows);  // kPlan* ids 0..4 ARE the AMP_DISC_PLAN_* values
  return AMP_OK;
}

static int style_reward_prescaled_impl(const AmpDisc* h, const void* xs_any, int64_t rows, float scale, const float* task, float task_w,
                                       float style_w, float* logits, float* style, float* combined, void* workspace,
                                       amp_stream_t stream, const CompactLaunch* compact);

int amp_disc_style_reward_prescaled(const AmpDisc* h, const void* xs_any, int64_t rows, float scale, const float* task, float task_w,
                                    float style_w, float* logits, float* style, float* combined, void* workspace,
                                    amp_stream_t stream) {
  return style_reward_prescaled_impl(h, xs_any, rows, scale, task, task_w, style_w, logits, style, combined, workspace, stream, nullptr);
}

int amp_disc_style_reward_prescaled_compact(const AmpDisc* h, const void* xs_any, int64_t rows, float scale, const float* task,
                                            float task_w, float style_w, float* logits, float* style, float* combined,
                                            void* workspace, const AmpCompactArgs* c, amp_stream_t stream) {
  AMP_REQUIRE(h && c, "amp_disc_style_reward_prescaled_compact: null argument");
  AMP_REQUIRE(c->num_envs >= 1 && c->mask && c->tile_counts && c->ids && c->count, "amp_disc_style_reward_prescaled_compact: null compaction buffer");
  AMP_REQUIRE(c->tile_envs == 8 || c->tile_envs == 16 || c->tile_envs == 32 || c->tile_envs == 64, "amp_disc_style_reward_prescaled_compact: tile_envs must be 8, 16, 32 or 64");
  AMP_REQUIRE(rows >= 1, "amp_disc_style_reward_prescaled_compact: needs at least one row");
  CompactLaunch cl;
  cl.mask = c->mask; cl.counts = c->tile_counts; cl.N = c->num_envs;
  cl.n_tiles = (c->num_envs + kTile - 1) / kTile;
  cl.n_counts = (c->num_envs + c->tile_envs - 1) / c->tile_envs;
  cl.sub = kTile / c->tile_envs; cl.ids = c->ids; cl.count = c->count;
  cl.blocks = (int)((cl.n_tiles + 3) / 4);
  if (!(h->mode == AMP_DISC_F16X3 && static_bound(h))) {
    // other engines / input formats: the two launches stay separate (same results)
    int rc = amp_reset_compact_tiles(c->mask, c->tile_counts, c->tile_envs, c->num_envs, c->ids, c->count, stream);
    if (rc != AMP_OK) return rc;
    return style_reward_prescaled_impl(h, xs_any, rows, scale, task, task_w, style_w, logits, style, combined, workspace, stream, nullptr);
  }
  return style_reward_prescaled_impl(h, xs_any, rows, scale, task, task_w, style_w, logits, style, combined, workspace, stream, &cl);
}

int amp_disc_style_reward_compact(const AmpDisc* h, const float* x, int64_t rows, int64_t row_stride, float scale, const float* task,
                                  float task_w, float style_w, float* logits, float* style, float* combined, void* workspace,
                                  const AmpCompactArgs* c, amp_stream_t stream) {
  AMP_REQUIRE(h && c && x && workspace, "amp_disc_style_reward_compact: null argument");
  AMP_REQUIRE(c->num_envs >= 1 && c->mask && c->tile_counts && c->ids && c->count, "amp_disc_style_reward_compact: null compaction buffer");
  AMP_REQUIRE(c->tile_envs == 8 || c->tile_envs == 16 || c->tile_envs == 32 || c->tile_envs == 64, "amp_disc_style_reward_compact: tile_envs must be 8, 16, 32 or 64");
  AMP_REQUIRE(rows >= 1 && rows <= ((int64_t)1 << 30), "amp_disc_style_reward_compact: rows out of range");
  AMP_REQUIRE(row_stride >= h->in_dim, "amp_disc_style_reward_compact: row_stride %lld < in_dim %d", (long long)row_stride, h->in_dim);
  AMP_REQUIRE((uintptr_t)workspace % 16 == 0, "amp_disc_style_reward_compact: workspace must be 16-byte aligned");
  if (!raw_rows_ok(h, rows, x, row_stride)) {
    // any other plan / engine: the two calls it stands for, back to back (same results)
    const int rc = amp_reset_compact_tiles(c->mask, c->tile_counts, c->tile_envs, c->num_envs, c->ids, c->count, stream);
    if (rc != AMP_OK) return rc;
    return amp_disc_style_reward(h, x, rows, row_stride, scale, task, task_w, style_w, logits, style, combined, workspace, nullptr, stream);
  }
  CompactLaunch cl;
  cl.mask = c->mask; cl.counts = c->tile_counts; cl.N = c->num_envs;
  cl.n_tiles = (c->num_envs + kTile - 1) / kTile;
  cl.n_counts = (c->num_envs + c->tile_envs - 1) / c->tile_envs;
  cl.sub = kTile / c->tile_envs; cl.ids = c->ids; cl.count = c->count;
  cl.blocks = (int)((cl.n_tiles + 3) / 4);
  const DiscWorkspace ws = disc_workspace(h, rows, workspace);
  const RawRows raw{x, row_stride};
  return disc_forward_f16(h, nullptr, nullptr, rows, (_Float16*)ws.h1, ws.partial, scale, task, task_w, style_w, logits, style, combined,
                          (hipStream_t)stream, &cl, &raw);
}

static int style_reward_prescaled_impl(const AmpDisc* h, const void* xs_any, int64_t rows, float scale, const float* task, float task_w,
                                       float style_w, float* logits, float* style, float* combined, void* workspace,
                                       amp_stream_t stream, const CompactLaunch* compact) {
  AMP_REQUIRE(h, "amp_disc_style_reward_prescaled: null handle");
  AMP_REQUIRE(rows >= 0, "amp_disc_style_reward_prescaled: negative rows");
  if (rows == 0) return AMP_OK;
  const float* xs = (const float*)xs_any;
  AMP_REQUIRE(xs && workspace, "amp_disc_style_reward_prescaled: null buffer");
  AMP_REQUIRE((uintptr_t)xs % 16 == 0 && (uintptr_t)workspace % 16 == 0, "amp_disc_style_reward_prescaled: 16-byte alignment required");
  AMP_REQUIRE(rows <= ((int64_t)1 << 30), "amp_disc_style_reward_prescaled: too many rows");
  const DiscWorkspace ws = disc_workspace(h, rows, workspace);
  hipStream_t st = (hipStream_t)stream;
  if (h->mode == AMP_DISC_F16X3 && static_bound(h))  // fp16 plane blocks at the clamp's plane scale (amp_disc_input_layout)
    return disc_forward_f16(h, (const _Float16*)xs_any, nullptr, rows, (_Float16*)ws.h1, ws.partial, scale, task, task_w, style_w,
                            logits, style, combined, st, compact);
  if (h->mode == AMP_DISC_F16X3) {
    // fp32 rows with no clamp to bound them: bound them by their abs-max, then split into planes
    AMP_HIP(hipMemsetAsync(ws.header, 0, sizeof(float), st));
    {
      amp::TraceScope trace__("disc_absmax_kernel", st);
      disc_absmax_kernel<<<1024, kBlock, 0, st>>>(xs, h->k1p, rows, h->in_dim, nullptr, nullptr, 0.0f, (unsigned*)ws.header);
    }
    {
      const int64_t quads = rows * (h->k1h / 4);
      amp::TraceScope trace__("disc_scale_split_kernel", st);
      disc_scale_split_kernel<<<(unsigned)((quads + kBlock - 1) / kBlock), kBlock, 0, st>>>(
          xs, h->k1p, rows, h->in_dim, h->k1h, nullptr, nullptr, 0.0f, h->range, ws.header, (uint32_t*)ws.xs, nullptr, nullptr);
    }
    int rcs = launch_status("disc_scale_split_kernel");
    if (rcs != AMP_OK) return rcs;
    return disc_forward_f16(h, (const _Float16*)ws.xs, ws.header, rows, (_Float16*)ws.h1, ws.partial, scale, task, task_w, style_w,
                            logits, style, combined, st);
  }
  return disc_forward(h, xs, rows, (float*)ws.h1, ws.partial, scale, task, task_w, style_w, logits, style, combined, st);
}

}  // extern "C"
