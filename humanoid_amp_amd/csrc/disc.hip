// Discriminator style reward: scaler -> Linear(K*D,1024)+ReLU -> Linear(1024,512)+ReLU -> Linear(512,1)
// -> -log(max(1 - sigmoid, 1e-4)) * scale -> reward mix.  fp32 end to end on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain; the 1e-5 budget rules out bf16/fp8 operands).
//
//   scale    one pass over amp_obs: RunningStandardScaler (exact fp32 divide, once per element) + zero padding of
//            K*D to a multiple of 32 -> Xs [M, k1p] in the workspace (16-B aligned rows for the GEMM staging).
//   layer 1  GEMM [M, k1p] x [k1p, 1024]: bias + ReLU in the epilogue, transposed through LDS so every store is
//            a full 256-B row segment; H1 [M,1024] written once to the workspace.
//   layer 2  GEMM [M,1024] x [1024,512]: bias + ReLU + the 512->1 output layer as an in-register dot with
//            w3, reduced over the tile's columns (lane butterfly, then LDS across the two column waves);
//            only per-(row, column-tile) partial logits leave the kernel.
//   finalize fixed-order sum of the 4 column-tile partials + b3, style reward, reward mix.
//
// Tile: 128 x 128 x 32, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA 32x32 accumulators (64 VGPRs).
// LDS rows are padded to 36 floats so the four ds_read_b128 a lane issues per operand row are conflict-free.
// Global->LDS staging is register-prefetched one k-tile ahead.  Workgroups are renumbered so that the
// column tiles of one row tile run on the same XCD (they share the A tile through that XCD's L2).
#include "disc_gemm.hpp"
#include "disc_gemm_split.hpp"

#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));  // native vector: HIP's f4 struct turns into memcpy -> scratch

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
  // opt-in split-precision GEMMs (disc_gemm_split.hpp): 0 = native fp32 MFMA, 2 = bf16x3, 3 = bf16x6
  int32_t planes;
  __bf16* w1s;  // [planes][h1][k1p]
  __bf16* w2s;  // [planes][h2][h1]
};

namespace amp {

constexpr int kMinN = 128;  // h1 / h2 must be multiples of the widest column tile
constexpr int kPadK = 16;  // in_dim is zero-padded to a multiple of the layer-1 k-tile

__global__ __launch_bounds__(kBlock) void disc_finalize_kernel(const float* __restrict__ partial, int n_tiles,
                                                               const float* __restrict__ b3, int64_t M, float scale,
                                                               const float* __restrict__ task, float task_w,
                                                               float style_w, float* __restrict__ logits,
                                                               float* __restrict__ style, float* __restrict__ combined) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= M) return;
  const float* p = partial + i * n_tiles;
  float s = 0.0f;
  if (n_tiles == 4) {
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
// split-precision forward on bf16 planes Xp [planes][rows][k1p] (disc_gemm_split.hpp)
static int disc_forward_split(const AmpDisc* h, const __bf16* Xp, int64_t rows, __bf16* H1p, float* partial, float scale,
                              const float* task, float task_w, float style_w, float* logits, float* style, float* combined,
                              hipStream_t st) {
  auto big_tiles = [&](int N) { return (rows + 127) / 128 * (N / 128) >= 512; };
  SplitGemmArgs g1{};
  g1.A = Xp; g1.a_plane = rows * h->k1p; g1.lda = h->k1p; g1.M = rows;
  g1.W = h->w1s; g1.w_plane = (int64_t)h->h1 * h->k1p; g1.Kp = h->k1p; g1.bias = h->b1; g1.N = h->h1;
  g1.C = H1p; g1.c_plane = rows * h->h1; g1.ldc = h->h1;
  {
    const bool big = big_tiles(h->h1);
    const int bm = big ? 128 : 64;
    g1.n_tiles = h->h1 / bm; g1.m_tiles = (int)((rows + bm - 1) / bm);
    const unsigned grid = (unsigned)(((int64_t)g1.m_tiles * g1.n_tiles + 7) / 8 * 8);
    amp::TraceScope trace__("disc_gemm_split_kernel<0>", st);
    if (h->planes == 3) {
      if (big) disc_gemm_split_kernel<128, 128, 3, 0, 2><<<grid, kBlock, 0, st>>>(g1);
      else disc_gemm_split_kernel<64, 64, 3, 0, 4><<<grid, kBlock, 0, st>>>(g1);
    } else {
      if (big) disc_gemm_split_kernel<128, 128, 2, 0, 2><<<grid, kBlock, 0, st>>>(g1);
      else disc_gemm_split_kernel<64, 64, 2, 0, 4><<<grid, kBlock, 0, st>>>(g1);
    }
  }
  int rc = launch_status("disc_gemm_split_kernel<0>");
  if (rc != AMP_OK) return rc;
  SplitGemmArgs g2{};
  g2.A = H1p; g2.a_plane = rows * h->h1; g2.lda = h->h1; g2.M = rows;
  g2.W = h->w2s; g2.w_plane = (int64_t)h->h2 * h->h1; g2.Kp = h->h1; g2.bias = h->b2; g2.N = h->h2;
  g2.w3 = h->w3; g2.partial = partial;
  {
    const bool big = big_tiles(h->h2);
    const int bm = big ? 128 : 64;
    g2.n_tiles = h->h2 / bm; g2.m_tiles = (int)((rows + bm - 1) / bm);
    const unsigned grid = (unsigned)(((int64_t)g2.m_tiles * g2.n_tiles + 7) / 8 * 8);
    amp::TraceScope trace__("disc_gemm_split_kernel<1>", st);
    if (h->planes == 3) {
      if (big) disc_gemm_split_kernel<128, 128, 3, 1, 2><<<grid, kBlock, 0, st>>>(g2);
      else disc_gemm_split_kernel<64, 64, 3, 1, 4><<<grid, kBlock, 0, st>>>(g2);
    } else {
      if (big) disc_gemm_split_kernel<128, 128, 2, 1, 2><<<grid, kBlock, 0, st>>>(g2);
      else disc_gemm_split_kernel<64, 64, 2, 1, 4><<<grid, kBlock, 0, st>>>(g2);
    }
  }
  rc = launch_status("disc_gemm_split_kernel<1>");
  if (rc != AMP_OK) return rc;
  { amp::TraceScope trace__("disc_finalize_kernel", st);
    disc_finalize_kernel<<<(unsigned)((rows + kBlock - 1) / kBlock), kBlock, 0, st>>>(partial, g2.n_tiles, h->b3, rows, scale, task,
                                                                                   task_w, style_w, logits, style, combined);
  }
  return launch_status("disc_finalize_kernel");
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
  // the fp32 weights changed in place: re-split the bf16 planes of the opt-in split-precision mode
  return h->planes ? amp_disc_set_precision(h, h->planes, (amp_stream_t)st) : AMP_OK;
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
  (void)hipFree(h->w1s);
  (void)hipFree(h->w2s);
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
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMalloc(&h->w1p, sizeof(float) * (size_t)h->h1 * h->k1p);
  if (e == hipSuccess) e = hipMalloc(&h->b1, sizeof(float) * h->h1);
  if (e == hipSuccess) e = hipMalloc(&h->w2, sizeof(float) * (size_t)h->h2 * h->h1);
  if (e == hipSuccess) e = hipMalloc(&h->b2, sizeof(float) * h->h2);
  if (e == hipSuccess) e = hipMalloc(&h->w3, sizeof(float) * h->h2);
  if (e == hipSuccess) e = hipMalloc(&h->b3, sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&h->mean, sizeof(float) * h->k1p);
  if (e == hipSuccess) e = hipMalloc(&h->den, sizeof(float) * h->k1p);
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
  if (rc == AMP_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(AMP_ERR_HIP, "amp_disc_create: stream sync failed");
  if (rc != AMP_OK) {
    amp_disc_destroy(h);
    return rc;
  }
  *out = h;
  return AMP_OK;
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
  h->clip = clip;
  h->has_scaler = true;
  return AMP_OK;
}

int amp_disc_set_precision(AmpDisc* h, int32_t bf16_planes, amp_stream_t stream) {
  AMP_REQUIRE(h, "amp_disc_set_precision: null handle");
  AMP_REQUIRE(bf16_planes == 0 || bf16_planes == 2 || bf16_planes == 3,
              "amp_disc_set_precision: planes must be 0 (native fp32), 2 (bf16x3) or 3 (bf16x6)");
  (void)hipFree(h->w1s);
  (void)hipFree(h->w2s);
  h->w1s = h->w2s = nullptr;
  h->planes = 0;
  if (bf16_planes == 0) return AMP_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n1 = (int64_t)h->h1 * h->k1p, n2 = (int64_t)h->h2 * h->h1;
  AMP_HIP(hipMalloc(&h->w1s, sizeof(__bf16) * n1 * bf16_planes));
  AMP_HIP(hipMalloc(&h->w2s, sizeof(__bf16) * n2 * bf16_planes));
  {
    amp::TraceScope trace__("disc_split_rows_kernel", st);
    disc_split_rows_kernel<<<(unsigned)((n1 / 4 + kBlock - 1) / kBlock), kBlock, 0, st>>>(h->w1p, h->k1p, h->h1, h->k1p, h->k1p, nullptr,
                                                                                        nullptr, 0.0f, bf16_planes, h->w1s, n1, nullptr, nullptr);
    disc_split_rows_kernel<<<(unsigned)((n2 / 4 + kBlock - 1) / kBlock), kBlock, 0, st>>>(h->w2, h->h1, h->h2, h->h1, h->h1, nullptr,
                                                                                        nullptr, 0.0f, bf16_planes, h->w2s, n2, nullptr, nullptr);
  }
  int rc = launch_status("disc_split_rows_kernel");
  if (rc != AMP_OK) return rc;
  h->planes = bf16_planes;
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

int64_t amp_disc_workspace_bytes(const AmpDisc* h, int64_t rows) {
  if (!h || rows < 0) return -1;
  // fp32 path: Xs, H1 as fp32; split path: 3 bf16 planes each (the same offsets are used by both)
  const int64_t elt = h->planes ? 6 : 4;
  const int64_t xs_bytes = round_up(elt * rows * h->k1p, 256);
  const int64_t h1_bytes = round_up(elt * rows * h->h1, 256);
  const int64_t part_bytes = round_up((int64_t)sizeof(float) * rows * (h->h2 / 64), 256);
  const int64_t task_bytes = round_up((int64_t)sizeof(float) * rows, 256);
  return xs_bytes + h1_bytes + part_bytes + task_bytes;
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
  const int64_t elt = h->planes ? 6 : 4;
  float* Xs = (float*)workspace;
  float* H1 = (float*)((char*)Xs + round_up(elt * rows * h->k1p, 256));
  float* partial = (float*)((char*)H1 + round_up(elt * rows * h->h1, 256));
  float* task_copy = (float*)((char*)partial + round_up((int64_t)sizeof(float) * rows * (h->h2 / 64), 256));
  if (h->planes) {
    // split-precision path: scaler + split into bf16 planes in one pass (also snapshots the task reward)
    __bf16* Xp = (__bf16*)Xs;
    __bf16* H1p = (__bf16*)H1;
    {
      const int64_t quads = rows * (h->k1p / 4);
      amp::TraceScope trace__("disc_split_rows_kernel", st);
      disc_split_rows_kernel<<<(unsigned)((quads + kBlock - 1) / kBlock), kBlock, 0, st>>>(
          x, row_stride, rows, h->in_dim, h->k1p, h->has_scaler ? h->mean : nullptr, h->den, h->clip, h->planes, Xp,
          rows * h->k1p, task, task_copy);
    }
    int rcs = launch_status("disc_split_rows_kernel");
    if (rcs != AMP_OK) return rcs;
    if (inputs_consumed) AMP_HIP(hipEventRecord((hipEvent_t)inputs_consumed, st));
    return disc_forward_split(h, Xp, rows, H1p, partial, scale, task ? task_copy : nullptr, task_w, style_w, logits, style,
                              combined, st);
  }
  {
    const int64_t quads = rows * (h->k1p / 4);
    amp::TraceScope trace__("disc_scale_pad_kernel", st);
    disc_scale_pad_kernel<<<(unsigned)((quads + kBlock - 1) / kBlock), kBlock, 0, st>>>(
        x, row_stride, rows, h->in_dim, h->k1p, h->has_scaler ? h->mean : nullptr, h->den, h->clip, Xs, task, task_copy);
  }
  int rc = launch_status("disc_scale_pad_kernel");
  if (rc != AMP_OK) return rc;
  // everything the caller handed in (amp_obs, task reward) has been consumed once this point of the stream is reached
  if (inputs_consumed) AMP_HIP(hipEventRecord((hipEvent_t)inputs_consumed, st));

  return disc_forward(h, Xs, rows, H1, partial, scale, task ? task_copy : nullptr, task_w, style_w, logits, style, combined, st);
}

int amp_disc_input_layout(const AmpDisc* h, int32_t* padded_dim, const float** mean, const float** den, float* clip) {
  AMP_REQUIRE(h, "amp_disc_input_layout: null handle");
  if (padded_dim) *padded_dim = h->k1p;
  if (mean) *mean = h->has_scaler ? h->mean : nullptr;
  if (den) *den = h->den;
  if (clip) *clip = h->clip;
  return AMP_OK;
}

int amp_disc_style_reward_prescaled(const AmpDisc* h, const float* xs, int64_t rows, float scale, const float* task, float task_w,
                                    float style_w, float* logits, float* style, float* combined, void* workspace,
                                    amp_stream_t stream) {
  AMP_REQUIRE(h, "amp_disc_style_reward_prescaled: null handle");
  AMP_REQUIRE(rows >= 0, "amp_disc_style_reward_prescaled: negative rows");
  if (rows == 0) return AMP_OK;
  AMP_REQUIRE(xs && workspace, "amp_disc_style_reward_prescaled: null buffer");
  AMP_REQUIRE((uintptr_t)xs % 16 == 0 && (uintptr_t)workspace % 16 == 0, "amp_disc_style_reward_prescaled: 16-byte alignment required");
  AMP_REQUIRE(rows <= ((int64_t)1 << 30), "amp_disc_style_reward_prescaled: too many rows");
  const int64_t elt = h->planes ? 6 : 4;
  float* H1 = (float*)((char*)workspace + round_up(elt * rows * h->k1p, 256));
  float* partial = (float*)((char*)H1 + round_up(elt * rows * h->h1, 256));
  hipStream_t st = (hipStream_t)stream;
  if (h->planes) {
    __bf16* Xp = (__bf16*)workspace;
    {
      const int64_t quads = rows * (h->k1p / 4);
      amp::TraceScope trace__("disc_split_rows_kernel", st);
      disc_split_rows_kernel<<<(unsigned)((quads + kBlock - 1) / kBlock), kBlock, 0, st>>>(
          xs, h->k1p, rows, h->k1p, h->k1p, nullptr, nullptr, 0.0f, h->planes, Xp, rows * h->k1p, nullptr, nullptr);
    }
    int rcs = launch_status("disc_split_rows_kernel");
    if (rcs != AMP_OK) return rcs;
    return disc_forward_split(h, Xp, rows, (__bf16*)H1, partial, scale, task, task_w, style_w, logits, style, combined, st);
  }
  return disc_forward(h, xs, rows, H1, partial, scale, task, task_w, style_w, logits, style, combined, st);
}


}  // extern "C"
