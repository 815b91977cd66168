// Discriminator style reward: scaler -> Linear(K*D,1024)+ReLU -> Linear(1024,512)+ReLU -> Linear(512,1)
// -> -log(max(1 - sigmoid, 1e-4)) * scale -> reward mix.  fp32 end to end on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain; the 1e-5 budget rules out bf16/fp8 operands).
//
//   layer 1  GEMM [M, in] x [in, 1024]: the running-standard-scaler is applied while the A tile is staged,
//            bias + ReLU in the epilogue, H1 [M,1024] written once to the workspace.
//   layer 2  GEMM [M,1024] x [1024,512]: bias + ReLU + the 512->1 output layer as an in-register dot with
//            w3, reduced over the tile's columns (lane butterfly, then LDS across the two column waves);
//            only per-(row, column-tile) partial logits leave the kernel.
//   finalize fixed-order sum of the 4 column-tile partials + b3, style reward, reward mix.
//
// Tile: 128 x 128 x 32, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA 32x32 accumulators (64 VGPRs).
// LDS rows are padded to 36 floats so the four ds_read_b128 a lane issues per operand row are conflict-free.
// Global->LDS staging is register-prefetched one k-tile ahead.  Workgroups are renumbered so that the
// column tiles of one row tile run on the same XCD (they share the A tile through that XCD's L2).
#include "amp_common.hpp"

typedef float floatx16 __attribute__((ext_vector_type(16)));
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
};

namespace amp {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;

struct GemmArgs {
  const float* A; int64_t lda; int64_t M; int32_t K;   // K = valid columns of A
  const float* W; int32_t Kp;                          // W [N, Kp], Kp % BK == 0
  const float* bias; int32_t N;
  const float* mean; const float* den; float clip;     // scaler (mode 0, optional)
  float* C; int64_t ldc;                               // mode 0 output
  const float* w3; float* partial; int32_t n_tiles;    // mode 1 output [M, n_tiles]
  int32_t m_tiles;
};

// XCD-aware renumbering: hardware deals workgroups round-robin over the 8 XCDs; give each XCD a contiguous
// run of (row tile, column tile) pairs with the column tile fastest (speed only, never correctness).
__device__ __forceinline__ bool tile_of_block(const GemmArgs& g, int& mt, int& nt) {
  const int total = g.m_tiles * g.n_tiles;
  const int per_xcd = (total + 7) / 8;
  const int v = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (v >= total) return false;
  mt = v / g.n_tiles;
  nt = v - mt * g.n_tiles;
  return true;
}


template <int MODE>
struct Stage;

// weights tile: every lane carries four f4 (rows r4 + 32*i of the [128, 32] tile)
struct StageB {
  f4 b0, b1, b2, b3;
  __device__ __forceinline__ void load_b(const GemmArgs& g, int n0, int kt, int c4, int r4) {
    const float* w = g.W + (int64_t)(n0 + r4) * g.Kp + kt * BK + 4 * c4;
    const int64_t step = (int64_t)32 * g.Kp;
    b0 = *reinterpret_cast<const f4*>(w);
    b1 = *reinterpret_cast<const f4*>(w + step);
    b2 = *reinterpret_cast<const f4*>(w + 2 * step);
    b3 = *reinterpret_cast<const f4*>(w + 3 * step);
  }
  __device__ __forceinline__ void store_b(float* Bs, int c4, int r4) const {
    float* d = &Bs[r4 * LDT + 4 * c4];
    *reinterpret_cast<f4*>(d) = b0;
    *reinterpret_cast<f4*>(d + 32 * LDT) = b1;
    *reinterpret_cast<f4*>(d + 64 * LDT) = b2;
    *reinterpret_cast<f4*>(d + 96 * LDT) = b3;
  }
};

// layer 2: A = H1 [M, 1024], 16-B aligned rows -> f4 loads
template <>
struct Stage<1> : StageB {
  f4 a0, a1, a2, a3;
  __device__ __forceinline__ void load(const GemmArgs& g, int64_t m0, int n0, int kt, int c4, int r4, int, int) {
    load_b(g, n0, kt, c4, r4);
    const int64_t last = g.M - 1;
    const int64_t ma = m0 + r4, mb = ma + 32, mc = ma + 64, md = ma + 96;
    const float* base = g.A + kt * BK + 4 * c4;
    a0 = *reinterpret_cast<const f4*>(base + (ma < last ? ma : last) * g.lda);
    a1 = *reinterpret_cast<const f4*>(base + (mb < last ? mb : last) * g.lda);
    a2 = *reinterpret_cast<const f4*>(base + (mc < last ? mc : last) * g.lda);
    a3 = *reinterpret_cast<const f4*>(base + (md < last ? md : last) * g.lda);
  }
  __device__ __forceinline__ void store(float* As, float* Bs, int c4, int r4, int, int) const {
    store_b(Bs, c4, r4);
    float* d = &As[r4 * LDT + 4 * c4];
    *reinterpret_cast<f4*>(d) = a0;
    *reinterpret_cast<f4*>(d + 32 * LDT) = a1;
    *reinterpret_cast<f4*>(d + 64 * LDT) = a2;
    *reinterpret_cast<f4*>(d + 96 * LDT) = a3;
  }
};

// layer 1: A = amp_obs [M, in_dim] (rows only 8-B aligned, in_dim not a multiple of 32): scalar loads, k guard,
// and the skrl RunningStandardScaler  clamp((x - mean) / (sqrt(var) + eps), -clip, clip)  applied in flight
template <>
struct Stage<0> : StageB {
  floatx16 a;
  __device__ __forceinline__ void load(const GemmArgs& g, int64_t m0, int n0, int kt, int c4, int r4, int k1, int r1) {
    load_b(g, n0, kt, c4, r4);
    const int k = kt * BK + k1;
    const bool kin = k < g.K;
    const bool scale = g.mean != nullptr;
    const float mu = (kin && scale) ? g.mean[k] : 0.0f;
    const float dn = (kin && scale) ? g.den[k] : 1.0f;
    const int64_t last = g.M - 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int64_t m = m0 + r1 + 8 * i;
      m = m < last ? m : last;
      float x = kin ? g.A[m * g.lda + k] : 0.0f;
      if (scale) {
        x = (x - mu) / dn;
        x = fminf(fmaxf(x, -g.clip), g.clip);
      }
      a[i] = kin ? x : 0.0f;
    }
  }
  __device__ __forceinline__ void store(float* As, float* Bs, int c4, int r4, int k1, int r1) const {
    store_b(Bs, c4, r4);
#pragma unroll
    for (int i = 0; i < 16; ++i) As[(r1 + 8 * i) * LDT + k1] = a[i];
  }
};

// One 128x128x32 tile step.  Lane (li, lh) takes k = 16*lh + s of operand row li for MFMA step s: any fixed
// permutation of k is a valid summation order as long as A and B use the same one, and this one turns the
// operand fetch into four ds_read_b128 per 32-row fragment.
__device__ __forceinline__ void compute_tile(const float* As, const float* Bs, int wm, int wn, int li, int lh,
                                             floatx16 (&acc)[2][2]) {
  const f4* pa0 = reinterpret_cast<const f4*>(&As[(wm * 64 + li) * LDT + 16 * lh]);
  const f4* pb0 = reinterpret_cast<const f4*>(&Bs[(wn * 64 + li) * LDT + 16 * lh]);
  const f4* pa1 = pa0 + 32 * LDT / 4;
  const f4* pb1 = pb0 + 32 * LDT / 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f4 x0 = pa0[q], x1 = pa1[q], y0 = pb0[q], y1 = pb1[q];
#define AMP_MFMA_STEP(c)                                                                   \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.c, y0.c, acc[0][0], 0, 0, 0); \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.c, y1.c, acc[0][1], 0, 0, 0); \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.c, y0.c, acc[1][0], 0, 0, 0); \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.c, y1.c, acc[1][1], 0, 0, 0);
    AMP_MFMA_STEP(x)
    AMP_MFMA_STEP(y)
    AMP_MFMA_STEP(z)
    AMP_MFMA_STEP(w)
#undef AMP_MFMA_STEP
  }
}

template <int MODE>
__global__ __launch_bounds__(kBlock, 2) void disc_gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
  int mt, nt;
  if (!tile_of_block(g, mt, nt)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)mt * BM;
  const int n0 = nt * BN;
  const int nk = g.Kp / BK;

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // ---- staging registers (named scalars, not arrays: keeps them out of scratch) -------------------
  const int c4 = tid & 7, r4 = tid >> 3;     // f4 mapping: row r4 + 32*i, floats [4*c4, 4*c4+4)
  const int k1 = tid & 31, r1 = tid >> 5;    // scalar mapping: row r1 + 8*i, float k1
  Stage<MODE> stg;
  stg.load(g, m0, n0, 0, c4, r4, k1, r1);
  stg.store(As, Bs, c4, r4, k1, r1);
  __syncthreads();
  for (int kt = 0; kt + 1 < nk; ++kt) {
    stg.load(g, m0, n0, kt + 1, c4, r4, k1, r1);  // in flight under this tile's MFMAs
    compute_tile(As, Bs, wm, wn, li, lh, acc);
    __syncthreads();
    stg.store(As, Bs, c4, r4, k1, r1);
    __syncthreads();
  }
  compute_tile(As, Bs, wm, wn, li, lh, acc);
  __syncthreads();

  // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) --------
  if (MODE == 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int col = n0 + wn * 64 + b * 32 + li;
        const float bias = g.bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < g.M) g.C[row * g.ldc + col] = fmaxf(acc[a][b][r] + bias, 0.0f);
        }
      }
  } else {
    // relu(acc + b2) . w3 over this tile's 128 columns
    float* red = As;  // [2][128] reuse (all waves passed the last barrier of the k loop)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float part[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) part[r] = 0.0f;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int col = n0 + wn * 64 + b * 32 + li;
        const float bias = g.bias[col], w = g.w3[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] += fmaxf(acc[a][b][r] + bias, 0.0f) * w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = part[r];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (li == 0) red[wn * BM + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = v;
      }
    }
    __syncthreads();
    if (tid < BM) {
      const int64_t row = m0 + tid;
      if (row < g.M) g.partial[row * g.n_tiles + nt] = red[tid] + red[BM + tid];
    }
  }
}

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

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

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
  delete h;
  return AMP_OK;
}

int amp_disc_create(const AmpDiscDesc* d, amp_stream_t stream, AmpDisc** out) {
  AMP_REQUIRE(d && out, "amp_disc_create: null argument");
  AMP_REQUIRE(d->in_dim >= 1, "amp_disc_create: in_dim must be positive");
  AMP_REQUIRE(d->h1 >= BN && d->h1 % BN == 0 && d->h1 % BK == 0, "amp_disc_create: h1=%d must be a multiple of %d", d->h1, BN);
  AMP_REQUIRE(d->h2 >= BN && d->h2 % BN == 0, "amp_disc_create: h2=%d must be a multiple of %d", d->h2, BN);
  AMP_REQUIRE(d->w1 && d->b1 && d->w2 && d->b2 && d->w3 && d->b3, "amp_disc_create: null weight pointer");
  AmpDisc* h = new (std::nothrow) AmpDisc();
  AMP_REQUIRE(h, "amp_disc_create: out of host memory");
  *h = AmpDisc{};
  h->in_dim = d->in_dim;
  h->h1 = d->h1;
  h->h2 = d->h2;
  h->k1p = (int32_t)round_up(d->in_dim, BK);
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

int64_t amp_disc_workspace_bytes(const AmpDisc* h, int64_t rows) {
  if (!h || rows < 0) return -1;
  const int64_t h1_bytes = round_up((int64_t)sizeof(float) * rows * h->h1, 256);
  const int64_t part_bytes = round_up((int64_t)sizeof(float) * rows * (h->h2 / BN), 256);
  return h1_bytes + part_bytes;
}

int amp_disc_style_reward(const AmpDisc* h, const float* x, int64_t rows, int64_t row_stride, float scale, const float* task,
                          float task_w, float style_w, float* logits, float* style, float* combined, void* workspace,
                          amp_stream_t stream) {
  AMP_REQUIRE(h, "amp_disc_style_reward: null handle");
  AMP_REQUIRE(rows >= 0, "amp_disc_style_reward: negative rows");
  if (rows == 0) return AMP_OK;
  AMP_REQUIRE(x && workspace, "amp_disc_style_reward: null buffer");
  AMP_REQUIRE(row_stride >= h->in_dim, "amp_disc_style_reward: row_stride %lld < in_dim %d", (long long)row_stride, h->in_dim);
  AMP_REQUIRE((uintptr_t)workspace % 16 == 0, "amp_disc_style_reward: workspace must be 16-byte aligned");
  AMP_REQUIRE(rows <= ((int64_t)1 << 24) * BM, "amp_disc_style_reward: too many rows");
  hipStream_t st = (hipStream_t)stream;
  float* H1 = (float*)workspace;
  float* partial = (float*)((char*)workspace + round_up((int64_t)sizeof(float) * rows * h->h1, 256));
  const int m_tiles = (int)((rows + BM - 1) / BM);

  GemmArgs g1{};
  g1.A = x; g1.lda = row_stride; g1.M = rows; g1.K = h->in_dim;
  g1.W = h->w1p; g1.Kp = h->k1p; g1.bias = h->b1; g1.N = h->h1;
  g1.mean = h->has_scaler ? h->mean : nullptr; g1.den = h->den; g1.clip = h->clip;
  g1.C = H1; g1.ldc = h->h1; g1.n_tiles = h->h1 / BN; g1.m_tiles = m_tiles;
  const unsigned grid1 = (unsigned)(((int64_t)m_tiles * g1.n_tiles + 7) / 8 * 8);
  { amp::TraceScope trace__("disc_gemm_kernel<0>", st);
    disc_gemm_kernel<0><<<grid1, kBlock, 0, st>>>(g1);
  }
  int rc = launch_status("disc_gemm_kernel<0>");
  if (rc != AMP_OK) return rc;

  GemmArgs g2{};
  g2.A = H1; g2.lda = h->h1; g2.M = rows; g2.K = h->h1;
  g2.W = h->w2; g2.Kp = h->h1; g2.bias = h->b2; g2.N = h->h2;
  g2.w3 = h->w3; g2.partial = partial; g2.n_tiles = h->h2 / BN; g2.m_tiles = m_tiles;
  const unsigned grid2 = (unsigned)(((int64_t)m_tiles * g2.n_tiles + 7) / 8 * 8);
  { amp::TraceScope trace__("disc_gemm_kernel<1>", st);
    disc_gemm_kernel<1><<<grid2, kBlock, 0, st>>>(g2);
  }
  rc = launch_status("disc_gemm_kernel<1>");
  if (rc != AMP_OK) return rc;

  { amp::TraceScope trace__("disc_finalize_kernel", st);
    disc_finalize_kernel<<<(unsigned)((rows + kBlock - 1) / kBlock), kBlock, 0, st>>>(partial, g2.n_tiles, h->b3, rows, scale, task,
                                                                                   task_w, style_w, logits, style, combined);
  }
  return launch_status("disc_finalize_kernel");
}

}  // extern "C"
