// Discriminator TRAINING step: the discriminator part of skrl's AMP._update, restated
// [skrl is third-party and absent: parity unpinned; oracle = torch autograd, oracle/disc_train.py].
//
//   loss = loss_scale * ( 0.5 * [BCE(D(policy U replay), 0) + BCE(D(motion), 1)]
//                         + logit_reg * |w3|^2 + grad_penalty * mean_rows |dD/dx (motion)|^2 + weight_decay * sum |W|^2 )
//   followed by one Adam step on (W1, b1, W2, b2, w3, b3).
//
// Everything is fp32 on the fp32 MFMA GEMM of disc_gemm.hpp.  That kernel is "NT" (both operands K-contiguous), so
// products that reduce over the batch dimension (dW = dY^T X) run on explicitly transposed copies; the batches are
// small (3 x 4096 rows), the transposes are noise.  ReLU masks are read back from the stored activations (H > 0).
// The gradient penalty's second-order term is analytic: with the masks m1, m2 fixed,
//   a2 = m2 * w3,  a1 = m1 * (a2 W2),  g = a1 W1,  P = mean |g|^2,  dg = 2 g / B
//   dW1 += a1^T dg,  e1 = m1 * (dg W1^T),  dW2 += a2^T e1,  dw3 += colsum(m2 * (e1 W2^T)).
#include "disc_gemm.hpp"
#include <cstdlib>
#include "disc_gemm_f16_dma.hpp"

struct AmpDisc;  // defined in disc.hip; accessed through the accessors below

namespace amp {
// accessors implemented in disc.hip
struct DiscParams {
  int32_t in_dim, h1, h2, k1p;
  float *w1p, *b1, *w2, *b2, *w3, *b3;
};
DiscParams disc_params(AmpDisc* h);
int disc_refresh_derived(AmpDisc* h, hipStream_t st);  // after the weights changed (split planes, ...)
int disc_range_reset(AmpDisc* h, hipStream_t st);      // zero the range record's accumulators / ticket (off the hot path)

static inline int64_t up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---- small kernels ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void transpose_pad_kernel(const float* __restrict__ in, int64_t rows, int cols, int64_t ld_in,
                                                               float* __restrict__ out, int64_t ld_out, int out_rows) {
  // out[c][r] = in[r][c] for c < cols, r < rows; zero for c in [cols, out_rows) and r in [rows, ld_out)
  __shared__ float tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int64_t r = r0 + i;
    const int c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[r * ld_in + c] : 0.0f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i;
    const int64_t r = r0 + tx;
    if (c < out_rows && r < ld_out) out[(int64_t)c * ld_out + r] = tile[tx][i];
  }
}

// Column sums in two deterministic stages.  Stage 1: grid (ceil(cols/64), kChunks); block (x, ch) sums its row chunk of
// 64 columns -> part[ch][c].  Stage 2: out[c] (+)= sum over ch in order.
// value(i, c) = A[i][c] * (rowscale ? rowscale[i] : 1) * (mask ? mask[i][c] > 0 : 1)
constexpr int kChunks = 64;
__global__ __launch_bounds__(kBlock) void colsum_part_kernel(const float* __restrict__ A, int64_t rows, int cols, int64_t lda,
                                                             const float* __restrict__ rowscale, const float* __restrict__ mask,
                                                             int64_t ldmask, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), p = threadIdx.x >> 6;
  const int64_t per = (rows + kChunks - 1) / kChunks;
  const int64_t lo = (int64_t)blockIdx.y * per, hi = lo + per < rows ? lo + per : rows;
  float s = 0.0f;
  if (c < cols)
    for (int64_t i = lo + p; i < hi; i += 4) {
      float v = A[i * lda + c];
      if (rowscale) v *= rowscale[i];
      if (mask) v = mask[i * ldmask + c] > 0.0f ? v : 0.0f;
      s += v;
    }
  red[p][threadIdx.x & 63] = s;
  __syncthreads();
  if (p == 0 && c < cols)
    part[(int64_t)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(kBlock) void colsum_final_kernel(const float* __restrict__ part, int cols, float* __restrict__ out,
                                                              int accumulate) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= cols) return;
  float s = 0.0f;
  for (int ch = 0; ch < kChunks; ++ch) s += part[(int64_t)ch * cols + c];
  out[c] = accumulate ? out[c] + s : s;
}

// out[e] (+)= sum over the split-K slices, in slice order (deterministic)
__global__ __launch_bounds__(kBlock) void sum_slices_kernel(const float* __restrict__ part, int slices, int64_t n, int64_t stride,
                                                            float* __restrict__ out, int accumulate) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  float s = accumulate ? out[e] : 0.0f;
  for (int k = 0; k < slices; ++k) s += part[(int64_t)k * stride + e];
  out[e] = s;
}

// deterministic scalar reduction tail: out[slot] (+)= scale * sum(part[0..n))
__global__ void scalar_final_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out, int slot,
                                    int accumulate) {
  __shared__ float red[256];
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[slot] = (accumulate ? out[slot] : 0.0f) + scale * red[0];
}

// one wave per row: logit[i] = A[i] . w + b, and -- the per-row half of bce_kernel, which was a single-block launch of its own --
// the row's BCE-with-logits term and dlogit[i]; bce_part[0 | 1][block] = the block's four rows' loss terms of the fake / motion
// group (summed in row order), reduced by the step's last kernel (reg_final_kernel).
__global__ __launch_bounds__(kBlock) void rowdot_bce_kernel(const float* __restrict__ A, int64_t rows, int cols, int64_t lda,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            int64_t n_fake, float loss_scale, float* __restrict__ out,
                                                            float* __restrict__ dlogit, float* __restrict__ bce_part) {
  __shared__ float s_l[2][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  float lf = 0.0f, lr = 0.0f;
  if (row < rows) {
    float s = 0.0f;
    for (int c = lane; c < cols; c += 64) s += A[row * lda + c] * w[c];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
      const float x = s + b[0];
      out[row] = x;
      const bool real = row >= n_fake;
      const float y = real ? 1.0f : 0.0f;
      // torch BCEWithLogits: (1 - y) x + max(-x, 0) + log(exp(-max(-x,0)) + exp(-x - max(-x,0)))
      const float mx = fmaxf(-x, 0.0f);
      const float l = (1.0f - y) * x + mx + logf(expf(-mx) + expf(-x - mx));
      const float sg = 1.0f / (1.0f + expf(-x));
      if (real) lr = l; else lf = l;
      dlogit[row] = loss_scale * 0.5f * (sg - y) / (float)(real ? rows - n_fake : n_fake);
    }
  }
  if (lane == 0) { s_l[0][wave] = lf; s_l[1][wave] = lr; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float* v = s_l[threadIdx.x];
    bce_part[threadIdx.x * gridDim.x + blockIdx.x] = ((v[0] + v[1]) + v[2]) + v[3];
  }
}

// dH2[i][n] = dlogit[i] * w3[n] * (H2[i][n] > 0)
__global__ __launch_bounds__(kBlock) void dh2_kernel(const float* __restrict__ dlogit, const float* __restrict__ w3,
                                                     const float* __restrict__ H2, int64_t rows, int cols,
                                                     float* __restrict__ dH2) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= rows * cols) return;
  const int64_t i = e / cols;
  const int n = (int)(e - i * cols);
  dH2[e] = H2[e] > 0.0f ? dlogit[i] * w3[n] : 0.0f;
}

// dh2_kernel + colsum(H2 * dlogit) + colsum(dH2) + colsum(dlogit) in ONE pass over H2 (they were seven launches reading H2 /
// dH2 three times).  Grid, row striding and the combination of the four row-lane partials are colsum_part_kernel's, so every
// sum is bit-identical to the separate launches:
//   dH2[i][c] = H2[i][c] > 0 ? dlogit[i] * w3[c] : 0
//   part[0][ch][c] = sum_i H2[i][c] * dlogit[i]   (-> gw3)      part[1][ch][c] = sum_i dH2[i][c]   (-> gb2)
//   part[2][ch][0] = sum_i dlogit[i]              (-> gb3; column block 0 only)
// `planes` (fp16-split step; cols % 32 == 0): also the two fp16 planes of plane_scale(bound[0]) * dH2 in block layout
// [rows][cols / 32][2][32] -- the A operand of dH1 = dH2 W2 on the fp16 pipe -- so that no separate split pass reads dH2 back.
__global__ __launch_bounds__(kBlock) void dh2_colsum_kernel(const float* __restrict__ dlogit, const float* __restrict__ w3,
                                                            const float* __restrict__ H2, int64_t rows, int cols,
                                                            float* __restrict__ dH2, float* __restrict__ part,
                                                            _Float16* __restrict__ planes, const float* __restrict__ bound) {
  __shared__ float red[3][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), p = threadIdx.x >> 6;
  const int64_t per = (rows + kChunks - 1) / kChunks;
  const int64_t lo = (int64_t)blockIdx.y * per, hi = lo + per < rows ? lo + per : rows;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
  const float w = c < cols ? w3[c] : 0.0f;
  const float sp = planes ? plane_scale(bound[0]) : 0.0f;
  _Float16* const pl = planes ? planes + (c >> 5) * 64 + (c & 31) : nullptr;
  if (c < cols)
    for (int64_t i = lo + p; i < hi; i += 4) {
      const float h = H2[i * cols + c], dl = dlogit[i];
      float v = h;
      v *= dl;                                    // colsum_part_kernel's "A * rowscale"
      s0 += v;
      const float d = h > 0.0f ? dl * w : 0.0f;   // dh2_kernel
      dH2[i * cols + c] = d;
      if (pl) {
        const float sd = d * sp;
        const _Float16 a = (_Float16)sd;
        pl[i * (2 * (int64_t)cols)] = a;
        pl[i * (2 * (int64_t)cols) + 32] = (_Float16)(sd - (float)a);
      }
      s1 += d;
      if (c == 0) s2 += dl;
    }
  red[0][p][threadIdx.x & 63] = s0;
  red[1][p][threadIdx.x & 63] = s1;
  red[2][p][threadIdx.x & 63] = s2;
  __syncthreads();
  if (p == 0 && c < cols) {
    const int l = threadIdx.x;
    const int64_t plane = (int64_t)kChunks * cols;
    part[0 * plane + (int64_t)blockIdx.y * cols + c] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
    part[1 * plane + (int64_t)blockIdx.y * cols + c] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
    if (c == 0) part[2 * plane + blockIdx.y] = (red[2][0][l] + red[2][1][l]) + (red[2][2][l] + red[2][3][l]);
  }
}
__global__ __launch_bounds__(kBlock) void dh2_colsum_final_kernel(const float* __restrict__ part, int cols, float* __restrict__ gw3,
                                                                  float* __restrict__ gb2, float* __restrict__ gb3) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  const int64_t plane = (int64_t)kChunks * cols;
  if (c < cols) {
    float a = 0.0f, b = 0.0f;
    for (int ch = 0; ch < kChunks; ++ch) { a += part[(int64_t)ch * cols + c]; b += part[plane + (int64_t)ch * cols + c]; }
    gw3[c] = a;
    gb2[c] = b;
  }
  if (c == 0) {
    float s = 0.0f;
    for (int ch = 0; ch < kChunks; ++ch) s += part[2 * plane + ch];
    gb3[0] = s;
  }
}

// a2[i][n] = w3[n] * (H2m[i][n] > 0)
__global__ __launch_bounds__(kBlock) void a2_kernel(const float* __restrict__ w3, const float* __restrict__ H2, int64_t rows, int cols,
                                                    float* __restrict__ a2) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= rows * cols) return;
  a2[e] = H2[e] > 0.0f ? w3[e % cols] : 0.0f;
}

// sum of squares of an [rows, cols] view (optionally scaling it in place by coef afterwards) -> per-block partials.
// A workgroup walks whole rows (row i of block b: b, b + grid, ...), its lanes the columns: no index division, and the four
// loads of an unrolled trip are in flight together (the flat-index form -- a 64-bit division and one dependent load per element,
// 52 elements per lane on 256 workgroups -- took 64 us on the [4096, 832] penalty gradient of K D = 830).
__global__ __launch_bounds__(kBlock) void sumsq_part_kernel(float* __restrict__ x, int64_t rows, int cols, int64_t ld, float coef,
                                                            int scale_in_place, float* __restrict__ part) {
  __shared__ float red[kBlock];
  float s = 0.0f;
  for (int64_t i = blockIdx.x; i < rows; i += gridDim.x) {
    float* row = x + i * ld;
    for (int c0 = threadIdx.x; c0 < cols; c0 += 4 * kBlock) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = c0 + j * kBlock < cols ? row[c0 + j * kBlock] : 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s += v[j] * v[j];
        if (scale_in_place && c0 + j * kBlock < cols) row[c0 + j * kBlock] = coef * v[j];
      }
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// Adam; the L2-type regularisers enter as grad += reg2 * p  (reg2 = 2 * loss_scale * coefficient).
// The six parameter tensors in ONE launch (they are 4-6 us each alone, launch-bound): segment s covers the flat indices
// [start[s], start[s + 1]).
// A segment's gradient may still be in pieces when Adam runs (the forked step): `slices[s]` split-K slices `slice_stride[s]` floats
// apart, summed here in slice order (what sum_slices_kernel did in a launch of its own), and, for segment 4 (w3), the
// gradient-penalty chain's column-sum partials `colpart[chunk * cols + c]` (colsum_final_kernel's accumulate).
struct AdamSegs {
  float* p[6]; const float* g[6]; float* m[6]; float* v[6];
  int64_t ld_p[6], ld_g[6], start[7];
  int64_t slice_stride[6];
  int32_t cols[6];
  int32_t slices[6];
  float reg2[6];
  const float* colpart;
  int32_t colpart_chunks;
};
// What changes from step to step besides the tensors lives on the DEVICE, so that a captured hipGraph of the training step
// advances it by itself: state[0] = samples the running scaler has seen (double), state[1] = Adam step (int64 bits),
// state[2] = the two bias corrections 1 - beta^step of the CURRENT step as floats (written by train_state_kernel, read by Adam).
struct TrainState {
  double count;
  long long step;
  float bc1, bc2;
};
__global__ void train_state_kernel(TrainState* __restrict__ s, double rows_merged, double beta1, double beta2) {
  s->count += rows_merged;
  s->step += 1;
  s->bc1 = (float)(1.0 - pow(beta1, (double)s->step));
  s->bc2 = (float)(1.0 - pow(beta2, (double)s->step));
}

// The kernel reads every parameter anyway, so it also leaves the sums of squares of W1 / W2 / w3 BEFORE the update (the reported
// regulariser terms; reg_sumsq_kernel was a separate 15-us launch): reg_part[k][block], k = 0 (W1), 1 (W2), 2 (w3).
__global__ __launch_bounds__(kBlock) void adam_multi_kernel(AdamSegs a, float lr, float b1, float b2, float eps,
                                                            const TrainState* __restrict__ state, float* __restrict__ grad_out,
                                                            float* __restrict__ reg_part) {
  __shared__ float red[3][kBlock];
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  float sq[3] = {0.0f, 0.0f, 0.0f};
  if (e < a.start[6]) {
  const float bc1 = state->bc1, bc2 = state->bc2;
  int s = 0;
#pragma unroll
  for (int k = 1; k < 6; ++k) s += e >= a.start[k];
  const int64_t i = e - a.start[s];
  const int64_t r = i / a.cols[s];
  const int c = (int)(i - r * a.cols[s]);
  float* pp = a.p[s] + r * a.ld_p[s] + c;
  const float pv = *pp;
  if (s == 0) sq[0] = pv * pv;
  if (s == 2) sq[1] = pv * pv;
  if (s == 4) sq[2] = pv * pv;
  const float* gp = a.g[s] + r * a.ld_g[s] + c;
  float gsum = gp[0];
  {  // the other slices, in slice order, sixteen loads in flight per trip (a wave's life in this kernel is a chain of memory round
     // trips: one dependent load per slice made it 35 us); a slot past the last slice adds +0.0f
    const int ns = a.slices[s];
    const int64_t ss = a.slice_stride[s];
    for (int k = 1; k < ns; k += 16) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = k + j < ns ? gp[(int64_t)(k + j) * ss] : 0.0f;
#pragma unroll
      for (int j = 0; j < 16; ++j) gsum += v[j];
    }
  }
  if (s == 4 && a.colpart != nullptr) {
    // kChunks (a constant: the loop unrolls, all loads in flight -- as a dynamic-length loop of dependent trips these two
    // workgroups were the kernel's whole 33 us); colpart_chunks == kChunks by construction
    float cs = 0.0f;
    const float* cp = a.colpart + c;
    const int nc = a.cols[4];
#pragma unroll
    for (int ch = 0; ch < kChunks; ++ch) cs += cp[ch * nc];
    gsum = gsum + cs;
  }
  const float gr = gsum + a.reg2[s] * pv;
  if (grad_out) grad_out[e] = gr;  // parameter order, logical shapes: the segments are laid out that way
  const float mi = b1 * a.m[s][i] + (1.0f - b1) * gr;
  const float vi = b2 * a.v[s][i] + (1.0f - b2) * gr * gr;
  a.m[s][i] = mi;
  a.v[s][i] = vi;
  *pp = pv - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  }
  if (reg_part == nullptr) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = sq[k];
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < 3) reg_part[threadIdx.x * gridDim.x + blockIdx.x] = red[threadIdx.x][0];
}

// loss[2] = logit_reg * sum w3^2 ; loss[3] = weight_decay * (sum W1^2 + sum W2^2 + sum w3^2)
// ... and loss[4] = loss_scale * (((loss[0] + loss[1]) + loss[2]) + loss[3]): the step's scaled total (skrl's discriminator_loss)
// ... and loss[0] = 0.5 * (mean BCE term of the fake rows + mean of the motion rows) from rowdot_bce_kernel's block partials.
// One block of 1 024 lanes: the five partial arrays (~2 700 + ~3 100 entries each) are two or three independent loads per lane
// (256 lanes walked them in 11 dependent trips: 15 us).
constexpr int kFinalBlock = 1024;
__global__ __launch_bounds__(kFinalBlock) void reg_final_kernel(const float* __restrict__ part, int n, float logit_reg,
                                                                float weight_decay, float loss_scale,
                                                                const float* __restrict__ bce_part, int n_bce, float n_fake,
                                                                float n_real, float* __restrict__ loss, float* __restrict__ loss_out) {
  __shared__ float red[5][kFinalBlock];
  float s[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = threadIdx.x; i < n; i += kFinalBlock) {
    s[0] += part[i];
    s[1] += part[n + i];
    s[2] += part[2 * n + i];
  }
  for (int i = threadIdx.x; i < n_bce; i += kFinalBlock) {
    s[3] += bce_part[i];
    s[4] += bce_part[n_bce + i];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int off = kFinalBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l0 = 0.5f * (red[3][0] / n_fake + red[4][0] / n_real);
    const float l2 = logit_reg * red[2][0];
    const float l3 = ((weight_decay * red[0][0]) + weight_decay * red[1][0]) + weight_decay * red[2][0];
    const float l1 = loss[1], l4 = loss_scale * (((l0 + l1) + l2) + l3);
    loss[0] = l0;
    loss[2] = l2;
    loss[3] = l3;
    loss[4] = l4;
    if (loss_out) {   // the caller's copy (was a 4.6-us device-to-device copy behind this kernel)
      loss_out[0] = l0; loss_out[1] = l1; loss_out[2] = l2; loss_out[3] = l3; loss_out[4] = l4;
    }
  }
}

// RunningStandardScaler update (train=True), two stages.  Stage 1: grid (ceil(cols/64), kChunks): per-chunk column sums
// of x and x^2 in fp64 (lanes = consecutive columns: coalesced).  Stage 2: batch mean / unbiased variance per column,
// merged into the running statistics (skrl _parallel_variance).
__global__ __launch_bounds__(kBlock) void scaler_part_kernel(const float* __restrict__ x, int64_t rows, int cols, int64_t ld,
                                                             double* __restrict__ part) {
  __shared__ double red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), p = threadIdx.x >> 6;
  const int64_t per = (rows + kChunks - 1) / kChunks;
  const int64_t lo = (int64_t)blockIdx.y * per, hi = lo + per < rows ? lo + per : rows;
  double s = 0.0, q = 0.0;
  if (c < cols) {
    // four rows' loads in flight per trip, summed in row order (one dependent load per trip made the kernel latency-bound)
    const float* px = x + c;
    int64_t i = lo + p;
    for (; i + 12 < hi; i += 16) {
      const float v0 = px[i * ld], v1 = px[(i + 4) * ld], v2 = px[(i + 8) * ld], v3 = px[(i + 12) * ld];
      s += (double)v0; q += (double)v0 * (double)v0;
      s += (double)v1; q += (double)v1 * (double)v1;
      s += (double)v2; q += (double)v2 * (double)v2;
      s += (double)v3; q += (double)v3 * (double)v3;
    }
    for (; i < hi; i += 4) {
      const double v = (double)px[i * ld];
      s += v;
      q += v * v;
    }
  }
  red[0][p][threadIdx.x & 63] = s;
  red[1][p][threadIdx.x & 63] = q;
  __syncthreads();
  if (p == 0 && c < cols) {
    const int l = threadIdx.x;
    part[((int64_t)blockIdx.y * cols + c) * 2 + 0] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
    part[((int64_t)blockIdx.y * cols + c) * 2 + 1] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
  }
}
// (+ the fp32 vectors the scaling pass reads -- mean32 / den32 [np], same formulas as disc_scaler_kernel -- so that a group's
//  batch is scaled with the statistics that include it without a round trip through the discriminator handle)
__global__ __launch_bounds__(kBlock) void scaler_merge_kernel(const double* __restrict__ part, int64_t rows, int cols,
                                                              double* __restrict__ mean, double* __restrict__ var,
                                                              const TrainState* __restrict__ state, double count_add,
                                                              int np, float eps, float* __restrict__ mean32,
                                                              float* __restrict__ den32) {
  // samples seen before this batch: the count at the start of the step (device state, advanced once at the step's end)
  // + the batches this step has merged already
  const double count = state->count + count_add;
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= cols) {
    if (mean32 && c < np) { mean32[c] = 0.0f; den32[c] = 1.0f; }
    return;
  }
  double s = 0.0, q = 0.0;
  for (int ch = 0; ch < kChunks; ++ch) {
    s += part[((int64_t)ch * cols + c) * 2 + 0];
    q += part[((int64_t)ch * cols + c) * 2 + 1];
  }
  const double n = (double)rows, bm = s / n;
  const double bv = (q - n * bm * bm) / (n - 1.0);  // fp64 sums of fp32 data: the cancellation is harmless
  const double total = count + n, delta = bm - mean[c];
  const double m2 = var[c] * count + bv * n + delta * delta * count * n / total;
  const double mean_new = mean[c] + delta * n / total, var_new = m2 / total;
  mean[c] = mean_new;
  var[c] = var_new;
  if (mean32) {
    mean32[c] = (float)mean_new;
    den32[c] = sqrtf((float)var_new) + eps;
  }
}

// The three batches (policy, replay, motion) of a step in ONE launch per stage (they were 3 x (partials, merge, scale)):
//   scaler_part3_kernel   blockIdx.z = batch: scaler_part_kernel's sums into part + z * stride
//   scaler_merge3_kernel  a column merges the three batches IN ORDER (each batch updates the running statistics and is then
//                         scaled with them, skrl's order) and keeps the fp32 vectors of every intermediate state
//   scale_rows3_kernel    blockIdx.y = batch: scaled with that batch's vectors
// Same arithmetic per column, in the same order, as the nine launches.
struct Batch3 {
  const float* x[3];
};
__global__ __launch_bounds__(kBlock) void scaler_part3_kernel(Batch3 b, int64_t rows, int cols, int64_t ld, double* __restrict__ part,
                                                              int64_t part_stride) {
  __shared__ double red[2][4][64];
  const float* __restrict__ x = b.x[blockIdx.z];
  part += (int64_t)blockIdx.z * part_stride;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), p = threadIdx.x >> 6;
  const int64_t per = (rows + kChunks - 1) / kChunks;
  const int64_t lo = (int64_t)blockIdx.y * per, hi = lo + per < rows ? lo + per : rows;
  double s = 0.0, q = 0.0;
  if (c < cols) {
    // four rows' loads in flight per trip, summed in row order (one dependent load per trip made the kernel latency-bound)
    const float* px = x + c;
    int64_t i = lo + p;
    for (; i + 12 < hi; i += 16) {
      const float v0 = px[i * ld], v1 = px[(i + 4) * ld], v2 = px[(i + 8) * ld], v3 = px[(i + 12) * ld];
      s += (double)v0; q += (double)v0 * (double)v0;
      s += (double)v1; q += (double)v1 * (double)v1;
      s += (double)v2; q += (double)v2 * (double)v2;
      s += (double)v3; q += (double)v3 * (double)v3;
    }
    for (; i < hi; i += 4) {
      const double v = (double)px[i * ld];
      s += v;
      q += v * v;
    }
  }
  red[0][p][threadIdx.x & 63] = s;
  red[1][p][threadIdx.x & 63] = q;
  __syncthreads();
  if (p == 0 && c < cols) {
    const int l = threadIdx.x;
    part[((int64_t)blockIdx.y * cols + c) * 2 + 0] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
    part[((int64_t)blockIdx.y * cols + c) * 2 + 1] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
  }
}
__global__ __launch_bounds__(kBlock) void scaler_merge3_kernel(const double* __restrict__ part, int64_t part_stride, int64_t rows,
                                                               int cols, double* __restrict__ mean, double* __restrict__ var,
                                                               const TrainState* __restrict__ state, int np, float eps,
                                                               float* __restrict__ mean32, float* __restrict__ den32, int vec_stride) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= cols) {
    if (mean32 && c < np)
      for (int g = 0; g < 3; ++g) { mean32[g * vec_stride + c] = 0.0f; den32[g * vec_stride + c] = 1.0f; }
    return;
  }
  double m = mean[c], v = var[c];
  const double n = (double)rows;
  // the three batches' sums first (independent of each other and of the merge below: their loads overlap), each in chunk order
  double sg[3], qg[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const double* pg = part + (int64_t)g * part_stride + (int64_t)c * 2;
    double s = 0.0, q = 0.0;
#pragma unroll 16
    for (int ch = 0; ch < kChunks; ++ch) {  // unrolled so that the loads are in flight together
      s += pg[(int64_t)ch * cols * 2 + 0];
      q += pg[(int64_t)ch * cols * 2 + 1];
    }
    sg[g] = s; qg[g] = q;
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const double count = state->count + (double)g * n;
    const double s = sg[g], q = qg[g];
    const double bm = s / n;
    const double bv = (q - n * bm * bm) / (n - 1.0);
    const double total = count + n, delta = bm - m;
    const double m2 = v * count + bv * n + delta * delta * count * n / total;
    m = m + delta * n / total;
    v = m2 / total;
    if (mean32) {
      mean32[g * vec_stride + c] = (float)m;
      den32[g * vec_stride + c] = sqrtf((float)v) + eps;
    }
  }
  mean[c] = m;
  var[c] = v;
}
// xs rows have a pitch of `pitch` floats (zero beyond k): batch z scaled with (mean + z * vec_stride, den + z * vec_stride)
// ones_col >= 0: that (padding) column is 1.0 in every row -- the TT product dH1^T Xs then leaves colsum(dH1), the bias gradient
// of layer 1, in column ones_col of gW1 for free (W1's padding columns are zero, so the forward GEMM does not see it)
__global__ __launch_bounds__(kBlock) void scale_rows3_kernel(Batch3 b, int64_t row_stride, int64_t rows, int k, int pitch,
                                                             const float* __restrict__ mean, const float* __restrict__ den,
                                                             int vec_stride, float clip, float* __restrict__ xs, int ones_col) {
  // one element per lane (consecutive lanes = consecutive floats of a row: a quad per lane with one 16-B store was slower, 32 vs
  // 23 us at K D = 830 -- the four 4-B loads of a lane are 16 B apart across the wave); 32-bit indices (rows * pitch < 2^31: host)
  const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= (uint32_t)rows * (uint32_t)pitch) return;
  const int z = blockIdx.y;
  const float* __restrict__ x = b.x[z];
  const uint32_t m = e / (uint32_t)pitch;
  const int c = (int)(e - m * (uint32_t)pitch);
  float v = 0.0f;
  if (c < k) {
    v = x[(int64_t)m * row_stride + c];
    if (mean) {
      v = (v - mean[z * vec_stride + c]) / den[z * vec_stride + c];
      v = fminf(fmaxf(v, -clip), clip);
    }
  } else if (c == ones_col) {
    v = 1.0f;
  }
  xs[(int64_t)z * rows * pitch + e] = v;
}

__global__ void scaler_to_f32_kernel(const double* __restrict__ mean64, const double* __restrict__ var64, int n, int np, float eps,
                                     float* __restrict__ mean, float* __restrict__ den) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  mean[i] = i < n ? (float)mean64[i] : 0.0f;
  den[i] = i < n ? sqrtf((float)var64[i]) + eps : 1.0f;
}

__global__ __launch_bounds__(kBlock) void scale_rows_kernel(const float* __restrict__ x, int64_t row_stride, int64_t rows, int k,
                                                            int kp, const float* __restrict__ mean, const float* __restrict__ den,
                                                            float clip, float* __restrict__ xs) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= rows * kp) return;
  const int64_t m = e / kp;
  const int c = (int)(e - m * kp);
  float v = 0.0f;
  if (c < k) {
    v = x[m * row_stride + c];
    if (mean) {
      v = (v - mean[c]) / den[c];
      v = fminf(fmaxf(v, -clip), clip);
    }
  }
  xs[e] = v;
}

}  // namespace amp

using namespace amp;

struct AmpDiscTrainer {
  AmpDisc* disc;
  AmpDiscTrainCfg cfg;
  DiscParams p;
  int64_t max_rows;   // capacity per group
  // persistent device state
  float *w2t, *w1t;              // W2^T [h1, h2], W1^T [kN, h1] (kN = k1p rounded to 64)
  float *mom[6], *vel[6];        // Adam moments for W1 (logical [h1, in_dim]), b1, W2, b2, w3, b3
  double *mean64, *var64;        // running scaler statistics (owned here when update_scaler)
  TrainState* state;             // device: scaler count, Adam step, bias corrections (see train_state_kernel)
  float* ws;                     // workspace
  int64_t ws_floats;
  int kN;
  // fp16-split GEMM path (cfg.gemm_f16x3): the two large products of the prediction loss's backward -- dH1 = (dH2 W2) * (H1 > 0) and
  // gW2 = dH2^T H1 -- run at fp32 accuracy on the fp16 matrix pipe (three MFMAs per product, disc_gemm_f16_dma.hpp MODE 2) on
  // block-layout planes written once per step, each with ONE power-of-two scale from a bound that costs no pass on the step's
  // critical path: dH2's from max|dlogit| x max|w3| (a one-workgroup kernel over 3 B + h2 floats), W2's and H1's from abs-max passes
  // on the side stream (under the scaler passes / the fp32 forward of layer 2).
  _Float16* dh2p;                // [Mp][h2 / 32][2][32]   planes of dH2            (K = h2: the A operand of dH1)
  _Float16* w2tp;                // [h1][h2 / 32][2][32]   planes of W2^T           (its W operand)
  _Float16* dh2tp;               // [h2][Mp / 32][2][32]   planes of dH2^T          (K = batch rows: the A operand of gW2)
  _Float16* h1tp;                // [h1][Mp / 32][2][32]   planes of H1^T           (its W operand)
  // ... and three of the gradient-penalty chain's six (a1 = a2 W2, gW2 += a2^T e1, da2 = e1 W2^T: 12.9 of its 17.7 GFLOP):
  _Float16* w2p;                 // [h2][h1 / 32][2][32]   planes of W2             (the W operand of da2)
  _Float16* a2p;                 // [Bp][h2 / 32][2][32]   planes of a2             (A of a1)
  _Float16* a2tp;                // [h2][Bp / 32][2][32]   planes of a2^T           (A of the penalty's gW2 product)
  _Float16* e1p;                 // [Bp][h1 / 32][2][32]   planes of e1             (A of da2)
  _Float16* e1tp;                // [h1][Bp / 32][2][32]   planes of e1^T           (W of the penalty's gW2 product)
  // ... and the prediction loss's W1 gradient gW1 = dH1^T Xs (output width kN padded to the 128-column tile: zero rows of Xs^T):
  _Float16* dh1tp;               // [h1][Mp / 32][2][32]   planes of dH1^T          (A of gW1)
  _Float16* xstp;                // [kNp][Mp / 32][2][32]  planes of Xs^T, kNp = kN rounded up to 128 (rows >= kN stay zero)
  // ... and the other three of the penalty chain (output / reduction width kN padded to kNp = kN rounded up to 128):
  _Float16* w1tp;                // [kNp][h1 / 32][2][32]  planes of W1^T (rows >= kN stay zero)      (W of g = a1 W1)
  _Float16* w1kp;                // [h1][kNp / 32][2][32]  planes of W1, k padded to kNp with zeros   (W of e1 = g W1^T)
  _Float16* a1p;                 // [Bp][h1 / 32][2][32]   planes of a1                               (A of g)
  _Float16* a1tp;                // [h1][Bp / 32][2][32]   planes of a1^T                             (A of gW1 += a1^T g)
  _Float16* gp;                  // [Bp][kNp / 32][2][32]  planes of g (after its in-place scaling)   (A of e1)
  _Float16* gtp;                 // [kNp][Bp / 32][2][32]  planes of g^T                              (W of gW1 += a1^T g)
  float* bound;                  // [16]: [0] bound of |dH2|, [1] max|W2|, [2] max|H1|, [3] max|w3| (bounds |a2|), [4] max|e1|,
                                 //      [5] h2 x [0] x [1] (bounds |dH1|), [6] max|Xs|, [8] max row sum of |W1|, [9] max|b1|; [2] = [8] x [6] + [9],
                                 //      [10] max|W1|, [11] h2 x [3] x [1] (bounds |a1|), [12] max|g|
  // the gradient-penalty chain (six GEMMs over the motion rows, grids that do not fill the chip) runs beside the prediction
  // loss's backward on a stream of the trainer's own: fork / join through these events (capturable: the side stream joins back)
  hipStream_t side;
  hipEvent_t ev[6];              // step start | W^T copies (+ W2^T planes) written | forward done | penalty chain done | H1 written | H1^T planes written
};

namespace {

unsigned blocks(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

constexpr int kShapeNotSupported = 1;  // "not for this shape" (not an error: the caller takes the fp32 engine)

// abs-max of a contiguous array of n4 float4 into out[0] (zeroed before): non-negative floats order like their bit patterns, so an
// integer atomicMax does it.  Four independent 16-B loads per lane and trip; a workgroup issues its atomic only if its maximum exceeds
// what the slot already holds (atomics on one address serialise in the L2 at ~50 ns each: after the first few workgroups almost
// none is needed).
__global__ __launch_bounds__(kBlock) void amax_flat_kernel(const fv4* __restrict__ x, int64_t n4, float* __restrict__ out) {
  __shared__ float s_part[kBlock / 64];
  float m = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n4; e += 4 * stride) {
    fv4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = e + j * stride < n4 ? x[e + j * stride] : fv4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[j][0]), fabsf(v[j][1]))), fmaxf(fabsf(v[j][2]), fabsf(v[j][3])));
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < kBlock / 64; ++i) m = fmaxf(m, s_part[i]);
    if (m > __builtin_nontemporal_load(out)) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
  }
}
// out[0] = max_j sum_k |W[j][k]| (zeroed before; one wave per row, conditional atomic as above), out[1] = max_j |b[j]|:
// |relu(x W^T + b)| <= out[0] * max|x| + out[1], the a-priori bound of the hidden layer (what DiscRange holds for inference)
__global__ __launch_bounds__(kBlock) void rowsum_bound_kernel(const float* __restrict__ W, int rows, int cols, int64_t ld,
                                                              const float* __restrict__ b, float* __restrict__ out) {
  __shared__ float s_part[2][kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * (kBlock / 64) + wave;
  float rs = 0.0f, bm = 0.0f;
  if (j < rows) {
    for (int k = lane; k < cols; k += 64) rs += fabsf(W[j * ld + k]);
    for (int off = 32; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
    bm = fabsf(b[j]);
  }
  if (lane == 0) { s_part[0][wave] = rs; s_part[1][wave] = bm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < kBlock / 64; ++i) { rs = fmaxf(rs, s_part[0][i]); bm = fmaxf(bm, s_part[1][i]); }
    if (rs > __builtin_nontemporal_load(out)) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(rs));
    if (bm > __builtin_nontemporal_load(out + 1)) atomicMax(reinterpret_cast<unsigned int*>(out + 1), __float_as_uint(bm));
  }
}
// out[0] = factor * a[0] * b[0] * 1.0001f + (c ? c[0] : 0)  (sums were rounded: keep the bound a bound)
__global__ void bound_affine_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                    float factor = 1.0f) {
  out[0] = factor * a[0] * b[0] * 1.0001f + (c ? c[0] : 0.0f);
}
// (rows x cols contiguous floats, rows * cols % 4 == 0)
void amax_flat(hipStream_t st, const float* x, int64_t n, float* out) {
  const int64_t n4 = n / 4;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, (n4 + 4 * kBlock - 1) / (4 * kBlock)));
  amax_flat_kernel<<<grid, kBlock, 0, st>>>(reinterpret_cast<const fv4*>(x), n4, out);
}

// out[0] = max_i |a[i]| * max_j |b[j]| (one workgroup of 1 024 lanes, twelve loads in flight per lane): the bound of
// |dH2| = |dlogit (x) w3 * mask| from 3 B + h2 floats
// (+ out2[0] = factor * out[0] * c[0]: the bound of the NEXT product's result)
__global__ __launch_bounds__(1024) void bound_product_kernel(const float* __restrict__ a, int64_t na, const float* __restrict__ b, int nb,
                                                             float* __restrict__ out, const float* __restrict__ c, float factor,
                                                             float* __restrict__ out2) {
  __shared__ float red[2][16];
  float ma = 0.0f, mb = 0.0f;
  for (int64_t i0 = threadIdx.x; i0 < na; i0 += 12 * 1024) {
    float v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) v[j] = i0 + (int64_t)j * 1024 < na ? a[i0 + (int64_t)j * 1024] : 0.0f;
#pragma unroll
    for (int j = 0; j < 12; ++j) ma = fmaxf(ma, fabsf(v[j]));
  }
  for (int i = threadIdx.x; i < nb; i += 1024) mb = fmaxf(mb, fabsf(b[i]));
  for (int off = 32; off > 0; off >>= 1) {
    ma = fmaxf(ma, __shfl_down(ma, off, 64));
    mb = fmaxf(mb, __shfl_down(mb, off, 64));
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ma; red[1][threadIdx.x >> 6] = mb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) { ma = fmaxf(ma, red[0][i]); mb = fmaxf(mb, red[1][i]); }
    out[0] = ma * mb;
    out2[0] = factor * (ma * mb) * c[0];
  }
}

// src [R, C] fp32 (row pitch ld) -> the planes of s * src^T in block layout dst[C][Rp / 32][2][32] halves (k = the source ROW index:
// the operand layout of a product that reduces over the batch), s = plane_scale(bound[0]); source rows in [R, Rp) are zero.
// One workgroup per 32 source rows x 64 columns through an LDS tile; a lane then owns 8 consecutive k of one column: one 16-B
// store per plane, four lanes fill a column's 128-B block.  C % 64 == 0, ld % 4 == 0, Rp % 32 == 0.
// `rows_dst` (optional): also the planes of s * src in ROW block layout [R][C / 32][2][32] (the operand layout of a product that
// reduces over the columns) from the same tile: one read of the source for both orientations.
__global__ __launch_bounds__(kBlock) void split_transpose_blocks_kernel(const float* __restrict__ src, int64_t R, int C, int64_t ld,
                                                                        const float* __restrict__ bound, _Float16* __restrict__ dst,
                                                                        int64_t Rp, _Float16* __restrict__ rows_dst = nullptr) {
  __shared__ float tile[32][65];
  const int c0 = blockIdx.x * 64;
  const int64_t r0 = (int64_t)blockIdx.y * 32;
  {
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;  // 16 column quads x 16 rows, two passes
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = rr + 16 * pass;
      fv4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (r0 + r < R) v = *reinterpret_cast<const fv4*>(src + (r0 + r) * ld + c0 + 4 * q);
      tile[r][4 * q] = v[0]; tile[r][4 * q + 1] = v[1]; tile[r][4 * q + 2] = v[2]; tile[r][4 * q + 3] = v[3];
    }
  }
  __syncthreads();
  const float s = plane_scale(bound[0]);
  const int col = threadIdx.x >> 2, part = threadIdx.x & 3;
  h8 p0, p1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = tile[8 * part + i][col] * s;
    const _Float16 a = (_Float16)v;
    p0[i] = a;
    p1[i] = (_Float16)(v - (float)a);
  }
  _Float16* out = dst + ((int64_t)(c0 + col) * (Rp >> 5) + blockIdx.y) * 64 + 8 * part;
  *reinterpret_cast<h8*>(out) = p0;
  *reinterpret_cast<h8*>(out + 32) = p1;
  if (rows_dst) {
    const int r = threadIdx.x >> 3, cg = threadIdx.x & 7;   // row of the tile, group of 8 columns (two k-blocks of four groups)
    if (r0 + r < R) {
      h8 q0, q1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = tile[r][8 * cg + i] * s;
        const _Float16 a = (_Float16)v;
        q0[i] = a;
        q1[i] = (_Float16)(v - (float)a);
      }
      _Float16* ro = rows_dst + (r0 + r) * (2 * (int64_t)C) + ((c0 >> 5) + (cg >> 2)) * 64 + 8 * (cg & 3);
      *reinterpret_cast<h8*>(ro) = q0;
      *reinterpret_cast<h8*>(ro + 32) = q1;
    }
  }
}

template <int TM, int TN>
int f16x3_kernel_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gemm_f16_dma_kernel<2, TM, TN>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, DmaTile<TM, TN>::kLds) == hipSuccess ? AMP_OK : AMP_ERR_HIP;
}
// the tiles' LDS exceeds the 64 KB default: the limit is raised once per device
int f16x3_init() {
  static bool done[64] = {};
  int dev = 0;
  AMP_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && done[dev]) return AMP_OK;
  if (f16x3_kernel_init<4, 2>() != AMP_OK || f16x3_kernel_init<4, 1>() != AMP_OK || f16x3_kernel_init<2, 1>() != AMP_OK)
    return fail(AMP_ERR_HIP, "amp_disc_trainer: raising the dynamic LDS limit of the fp16-split GEMM failed");
  if (dev >= 0 && dev < 64) done[dev] = true;
  return AMP_OK;
}

// Tile and k-slice plan of C[M, N] = A W^T on planes (K = 32 nq): 256 x 256 when that still gives ~3/4 of the CUs a workgroup, then
// 256 x 128, else 128 x 128; split-K (the weight-gradient product: a few output tiles, hundreds of k-blocks) on 128 x 128 tiles,
// at most `max_slices` slices of >= 8 k-blocks.
struct F16Plan { int tm, tn, slices; };
F16Plan f16_planes_plan(int64_t M, int N, int nq, int max_slices) {
  F16Plan q{2, 1, 1};
  auto tiles = [&](int bm, int bn) { return (int)((M + bm - 1) / bm) * (N / bn); };
  while (q.slices * 2 <= max_slices && tiles(128, 128) * q.slices * 2 <= 512 && nq % (q.slices * 2) == 0 && nq / (q.slices * 2) >= 8) q.slices *= 2;
  // The forked step runs these products BESIDE the other stream's fp32 GEMMs: a 256-row tile owns a CU's whole LDS and waits for
  // every co-resident workgroup of the other stream to drain before it can start; 128 x 128 tiles (two per CU) interleave
  // (K D = 166: 0.620 -> 0.593 ms per step, same box; AMP_TRAIN_F16_BIG=1 restores the large tiles for A/B runs)
  static const bool small_only = !(getenv("AMP_TRAIN_F16_BIG") && getenv("AMP_TRAIN_F16_BIG")[0] == '1');
  if (q.slices == 1 && !small_only) {
    if (N % 256 == 0 && tiles(256, 256) >= 192) { q.tm = 4; q.tn = 2; }
    else if (tiles(256, 128) >= 192) { q.tm = 4; q.tn = 1; }
  }
  return q;
}
// C[M, N] (ld = ldc) = A W^T at fp32 accuracy on the fp16 matrix pipe: A planes [M][Kp / 32][2][32] of plane_scale(bound_a[0]) A,
// W planes [N][Kp / 32][2][32] of plane_scale(bound_w[0]) W, the LDS-DMA kernel of the inference path (disc_gemm_f16_dma.hpp, MODE 2).
// plan.slices > 1: slice s of the reduction lands in C + s * slice_stride (summed by the caller: Adam does it for the weight gradients).
int gemm_f16_planes(hipStream_t st, const _Float16* A, int64_t M, const _Float16* W, int N, int Kp, const float* bound_a,
                    const float* bound_w, float* C, int64_t ldc, const float* mask, int64_t ldmask, F16Plan plan, int64_t slice_stride) {
  GemmF16Args g{};
  g.A = A; g.lda = Kp; g.M = M;
  g.W = W; g.Kp = Kp; g.N = N;
  g.C = C; g.ldc = ldc; g.amax_a = bound_a; g.amax_w = bound_w;
  g.mask = mask; g.ldmask = ldmask;
  const int bm = 64 * plan.tm, bn = 128 * plan.tn;
  g.m_tiles = (int)((M + bm - 1) / bm); g.n_tiles = N / bn;
  if (plan.slices > 1) { g.k_slices = plan.slices; g.slice_stride = slice_stride; }
  const unsigned grid = (unsigned)(((int64_t)g.m_tiles * g.n_tiles * plan.slices + 7) / 8 * 8);
  {
    amp::TraceScope trace__("disc_gemm_f16_dma_kernel<2>", st);
    if (plan.tm == 4 && plan.tn == 2) disc_gemm_f16_dma_kernel<2, 4, 2><<<grid, kDmaThreads, DmaTile<4, 2>::kLds, st>>>(g);
    else if (plan.tm == 4) disc_gemm_f16_dma_kernel<2, 4, 1><<<grid, kDmaThreads, DmaTile<4, 1>::kLds, st>>>(g);
    else disc_gemm_f16_dma_kernel<2, 2, 1><<<grid, kDmaThreads, DmaTile<2, 1>::kLds, st>>>(g);
  }
  return launch_status("disc_gemm_f16_dma_kernel<2>");
}

// C[M, N] (ld = ldc) = A W^T, optionally gated / accumulated.  `split` (scratch of k_slices * M * ldc floats) enables
// split-K for the weight-gradient products whose reduction runs over the 4096..12288 batch rows while the output is
// only a few hundred tiles: the slices' partial products land in `split` and are summed in slice order.
// 64 x 64 tiles of the backward NT products with 32-deep k-tiles: a 64 x 64 tile has only 8 MFMAs (512 matrix-pipe cycles) per wave and
// 16-deep k-tile between its two barriers and its LDS refill; 32-deep k-tiles halve the barriers per MFMA: 257-260 -> 246-248 us for the
// five products of a step (gpurun_out/r04_t; AMP_TRAIN_BK32=0 switches them off: A/B runs).
// The TT kernel adds its k-pairs in ascending order whatever the k-tile (bit-identical); the NT kernel's fixed k-permutation inside a
// k-tile (disc_gemm.hpp) follows the tile depth, i.e. the fp32 summation order changes within the parity bars of the training tests.
// AMP_TRAIN_FORK=0 keeps the whole step on the caller's stream (read per step: the tests compare the two in one process)
static bool fork_ok() {
  const char* e = getenv("AMP_TRAIN_FORK");
  return !(e && e[0] == '0');
}
static bool bk32_ok() {
  static const bool on = !(getenv("AMP_TRAIN_BK32") && getenv("AMP_TRAIN_BK32")[0] == '0');
  return on;
}
int gemm_nt(hipStream_t st, const float* A, int64_t lda, int64_t M, const float* W, int Kp, int N, float* C, int64_t ldc,
            const float* mask, int64_t ldmask, int accumulate, float* split = nullptr) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.M = M; g.K = Kp; g.W = W; g.Kp = Kp; g.N = N; g.C = C; g.ldc = ldc;
  g.mask = mask; g.ldmask = ldmask; g.accumulate = accumulate;
  // 128 x 128 tiles (0.8+ of the fp32 matrix peak) when they still give every CU two workgroups, else 64 x 64
  // (tools/gemm_bench.hip, profiles/r01_gemm_variants.txt); the element arithmetic does not depend on the tile
  const bool big = !split && N % 128 == 0 && (M + 127) / 128 * (N / 128) >= 512;
  const int bt = big ? 128 : 64;
  g.n_tiles = N / bt; g.m_tiles = (int)((M + bt - 1) / bt);
  const int tiles = g.m_tiles * g.n_tiles, nk = Kp / 16;
  int slices = 1;
  if (split && !mask) {
    while (slices < 16 && tiles * slices < 1024 && nk / (slices * 2) >= 16) slices *= 2;
  }
  if (slices > 1) {
    g.k_slices = slices; g.slice_stride = M * ldc; g.C = split; g.accumulate = 0;
  }
  const unsigned grid = (unsigned)(((int64_t)tiles * slices + 7) / 8 * 8);
  {
    amp::TraceScope trace__("disc_gemm_kernel<2>", st);
    if (big) disc_gemm_kernel<128, 128, 16, 1, 2, 4><<<grid, kBlock, 0, st>>>(g);
    else if (bk32_ok() && Kp % 32 == 0) disc_gemm_kernel<64, 64, 32, 1, 2, 4><<<grid, kBlock, 0, st>>>(g);
    else disc_gemm_kernel<64, 64, 16, 1, 2, 4><<<grid, kBlock, 0, st>>>(g);
  }
  int rc = launch_status("disc_gemm_kernel<2>");
  if (rc != AMP_OK || slices == 1) return rc;
  const int64_t n = M * ldc;
  sum_slices_kernel<<<(unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, st>>>(split, slices, n, n, C, accumulate);
  return launch_status("sum_slices_kernel");
}

// C[M, N] (ld = ldc) (+)= A^T W with A [K, M] (row pitch lda) and W [K, N] (row pitch ldw): the weight-gradient products
// dW = dY^T X straight on the row-major batch tensors (disc_gemm_tt_kernel: no transposed copies); split-K as gemm_nt.
// `defer`: the weight-gradient products of a weight come in pairs (prediction loss, gradient penalty): *defer == 0 on entry ->
// this call writes its k-slices into `split` and returns their number in *defer WITHOUT summing them; *defer > 0 on entry -> this
// call ACCUMULATES into the first min(*defer, own) slices, then the slices are summed once into C (one sum_slices launch and one
// pass over the slices per weight instead of two).
//
// Shape admission (tt_plan; also exported as amp_disc_train_tt_plan so that it is testable without a GPU).  The kernel has NO
// row / column guards on its tile -- its loads take BM_ / BN_ consecutive floats of every operand row and its epilogue stores BN_
// consecutive floats of BM_ rows of C -- so the host admits only shapes the chosen tile covers EXACTLY:
//   M % BM == 0 and N % BN == 0 for the tile that is launched (round 3 checked `% 64` whatever the tile: a 128-wide tile on
//   N = kN = 192 / 832 would have read and stored 64 columns past every row, profiles/r04_tt_kernel_abort.md),
//   lda >= M, ldw >= N, ldc >= N (the rows hold the tile), 16-B aligned rows (pitches % 4), K in int32.
// Tile choice: 128 x 128 (one LDS read per MFMA instead of two) or 128 x 64 where the shape divides AND tiles x k-slices still
// give every CU two workgroups (>= 512); 64 x 64 otherwise.  The element arithmetic (k ascending within a slice, slices summed in
// order) depends on the slice count only, not on the tile.
struct TtPlan { int bm, bn, slices; };
static int tt_slices(const int tiles, const int64_t K, const bool split) {
  const int nk = (int)((K + 15) / 16);
  int slices = 1;
  if (split)
    while (slices < 16 && tiles * slices < 1024 && nk / (slices * 2) >= 16) slices *= 2;
  return slices;
}
static int tt_plan(int M, int N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, bool split, TtPlan* out) {
  if (M <= 0 || N <= 0 || K <= 0 || K > INT32_MAX) return kShapeNotSupported;
  if (M % 64 != 0 || N % 64 != 0 || lda % 4 != 0 || ldw % 4 != 0 || ldc % 4 != 0) return kShapeNotSupported;
  if (lda < M || ldw < N || ldc < N) return kShapeNotSupported;
  // the slice count is the 64 x 64 plan's whatever tile runs: a row of C is then the same sum on every tile shape (bit-identical)
  const int slices = tt_slices((M / 64) * (N / 64), K, split);
  TtPlan p{64, 64, slices};
  // (the 512 re-measured under the forked step, where a second stream fills the chip: admitting the large tiles from 256 or 128
  //  workgroups made the step slower, 0.703-0.705 vs 0.680-0.687 ms, gpurun_out/r04_al)
  if (M % 128 == 0 && N % 128 == 0 && (M / 128) * (N / 128) * slices >= 512) p = TtPlan{128, 128, slices};
  else if (M % 128 == 0 && (M / 128) * (N / 64) * slices >= 512) p = TtPlan{128, 64, slices};
  if (M % p.bm != 0 || N % p.bn != 0) return kShapeNotSupported;  // (cannot happen by construction; the kernel's precondition)
  *out = p;
  return AMP_OK;
}

//
// `raw`: write this product's k-slices (even a single one) to `split` and return their number in *raw without summing -- the
// forked step's two products of a weight run on two streams, each into its own slice region, and one sum follows the join.
int gemm_tt(hipStream_t st, const float* A, int64_t lda, int M, const float* W, int64_t ldw, int N, int64_t K, float* C, int64_t ldc,
            int accumulate, float* split, int* defer = nullptr, int* raw = nullptr) {
  TtPlan plan;
  if (tt_plan(M, N, K, lda, ldw, ldc, split != nullptr, &plan) != AMP_OK) return kShapeNotSupported;
  if (raw) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.M = M; g.K = (int32_t)K; g.W = W; g.Kp = (int32_t)ldw; g.N = N; g.C = split; g.ldc = ldc;
    g.n_tiles = N / plan.bn; g.m_tiles = M / plan.bm;
    g.k_slices = plan.slices; g.slice_stride = (int64_t)M * ldc;
    const unsigned grid = (unsigned)(((int64_t)g.m_tiles * g.n_tiles * plan.slices + 7) / 8 * 8);
    {
      amp::TraceScope trace__("disc_gemm_tt_kernel", st);
      if (plan.bm == 128 && plan.bn == 128) disc_gemm_tt_kernel<128, 128, 16, 4><<<grid, kBlock, 0, st>>>(g);
      else if (plan.bm == 128) disc_gemm_tt_kernel<128, 64, 16, 4><<<grid, kBlock, 0, st>>>(g);
      else disc_gemm_tt_kernel<64, 64, 16, 4><<<grid, kBlock, 0, st>>>(g);
    }
    *raw = plan.slices;
    return launch_status("disc_gemm_tt_kernel");
  }
  GemmArgs g{};
  g.A = A; g.lda = lda; g.M = M; g.K = (int32_t)K; g.W = W; g.Kp = (int32_t)ldw; g.N = N; g.C = C; g.ldc = ldc;
  g.accumulate = accumulate;
  g.n_tiles = N / plan.bn; g.m_tiles = M / plan.bm;
  const int tiles = g.m_tiles * g.n_tiles;
  int slices = plan.slices;
  const int first = defer ? *defer : 0;  // slices the first product of the pair left in `split`
  if (first > 0 && slices > first) slices = first;
  const bool deferred = defer && slices > 1;
  if (slices > 1) {
    g.k_slices = slices; g.slice_stride = (int64_t)M * ldc; g.C = split; g.accumulate = first > 0 ? 1 : 0;
  } else if (first > 0) {
    // this product is not sliced: add it to slice 0 of the pair's first product
    g.C = split; g.accumulate = 1;
  }
  const unsigned grid = (unsigned)(((int64_t)tiles * slices + 7) / 8 * 8);
  {
    amp::TraceScope trace__("disc_gemm_tt_kernel", st);
    if (plan.bm == 128 && plan.bn == 128) disc_gemm_tt_kernel<128, 128, 16, 4><<<grid, kBlock, 0, st>>>(g);
    else if (plan.bm == 128) disc_gemm_tt_kernel<128, 64, 16, 4><<<grid, kBlock, 0, st>>>(g);
    else disc_gemm_tt_kernel<64, 64, 16, 4><<<grid, kBlock, 0, st>>>(g);  // (32-deep k-tiles: 257.1 vs 255.0 us per step, no gain)
  }
  int rc = launch_status("disc_gemm_tt_kernel");
  if (rc != AMP_OK) return rc;
  if (deferred && first == 0) { *defer = slices; return AMP_OK; }  // the pair's second product sums
  const int to_sum = first > 0 ? first : slices;
  if (to_sum == 1 && first == 0) return AMP_OK;
  const int64_t n = (int64_t)M * ldc;
  sum_slices_kernel<<<(unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, st>>>(split, to_sum, n, n, C, first > 0 ? 0 : accumulate);
  return launch_status("sum_slices_kernel");
}

int gemm_fwd(hipStream_t st, const float* A, int64_t lda, int64_t M, const float* W, int Kp, int N, const float* bias, float* C) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.M = M; g.K = Kp; g.W = W; g.Kp = Kp; g.bias = bias; g.N = N; g.C = C; g.ldc = N;
  const bool big = N % 128 == 0 && (M + 127) / 128 * (N / 128) >= 512;
  const int bt = big ? 128 : 64;
  g.n_tiles = N / bt; g.m_tiles = (int)((M + bt - 1) / bt);
  const unsigned grid = (unsigned)(((int64_t)g.m_tiles * g.n_tiles + 7) / 8 * 8);
  amp::TraceScope trace__("disc_gemm_kernel<0>", st);
  if (big) disc_gemm_kernel<128, 128, 16, 1, 0, 4><<<grid, kBlock, 0, st>>>(g);
  else disc_gemm_kernel<64, 64, 16, 1, 0, 8><<<grid, kBlock, 0, st>>>(g);  // (32-deep k-tiles: 162-166 vs 156-157 us for the two forwards: slower)
  return launch_status("disc_gemm_kernel<0>");
}

void transpose(hipStream_t st, const float* in, int64_t rows, int cols, int64_t ld_in, float* out, int64_t ld_out, int out_rows) {
  dim3 grid((unsigned)((ld_out + 31) / 32), (unsigned)((out_rows + 31) / 32));
  transpose_pad_kernel<<<grid, kBlock, 0, st>>>(in, rows, cols, ld_in, out, ld_out, out_rows);
}

}  // namespace

extern "C" {

int amp_disc_trainer_destroy(AmpDiscTrainer* t) {
  if (!t) return AMP_OK;
  (void)hipFree(t->w2t);
  (void)hipFree(t->w1t);
  for (int i = 0; i < 6; ++i) {
    (void)hipFree(t->mom[i]);
    (void)hipFree(t->vel[i]);
  }
  (void)hipFree(t->state);
  (void)hipFree(t->mean64);
  (void)hipFree(t->var64);
  (void)hipFree(t->ws);
  (void)hipFree(t->dh2p);
  (void)hipFree(t->w2tp);
  (void)hipFree(t->dh2tp);
  (void)hipFree(t->h1tp);
  (void)hipFree(t->w2p);
  (void)hipFree(t->w1tp);
  (void)hipFree(t->w1kp);
  (void)hipFree(t->a1p);
  (void)hipFree(t->a1tp);
  (void)hipFree(t->gp);
  (void)hipFree(t->gtp);
  (void)hipFree(t->dh1tp);
  (void)hipFree(t->xstp);
  (void)hipFree(t->a2p);
  (void)hipFree(t->a2tp);
  (void)hipFree(t->e1p);
  (void)hipFree(t->e1tp);
  (void)hipFree(t->bound);
  for (hipEvent_t e : t->ev)
    if (e) (void)hipEventDestroy(e);
  if (t->side) (void)hipStreamDestroy(t->side);
  delete t;
  return AMP_OK;
}

int amp_disc_trainer_create(AmpDisc* disc, const AmpDiscTrainCfg* cfg, const double* running_mean_dev,
                            const double* running_variance_dev, double current_count, amp_stream_t stream,
                            AmpDiscTrainer** out) {
  AMP_REQUIRE(disc && cfg && out, "amp_disc_trainer_create: null argument");
  AMP_REQUIRE(cfg->max_rows_per_group >= 16, "amp_disc_trainer_create: max_rows_per_group must be >= 16");
  AmpDiscTrainer* t = new (std::nothrow) AmpDiscTrainer();
  AMP_REQUIRE(t, "amp_disc_trainer_create: out of host memory");
  *t = AmpDiscTrainer{};
  t->disc = disc;
  t->cfg = *cfg;
  t->p = disc_params(disc);
  if (disc_range_reset(disc, (hipStream_t)stream) != AMP_OK) {
    delete t;
    return AMP_ERR_HIP;
  }
  t->max_rows = up(cfg->max_rows_per_group, 16);
  t->kN = (int)up(t->p.k1p, 64);
  const DiscParams& p = t->p;
  const int64_t sizes[6] = {(int64_t)p.h1 * p.in_dim, p.h1, (int64_t)p.h2 * p.h1, p.h2, p.h2, 1};
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMalloc(&t->w2t, sizeof(float) * (size_t)p.h1 * p.h2);
  if (e == hipSuccess) e = hipMalloc(&t->w1t, sizeof(float) * (size_t)t->kN * p.h1);
  for (int i = 0; i < 6 && e == hipSuccess; ++i) {
    e = hipMalloc(&t->mom[i], sizeof(float) * sizes[i]);
    if (e == hipSuccess) e = hipMalloc(&t->vel[i], sizeof(float) * sizes[i]);
    if (e == hipSuccess) e = hipMemsetAsync(t->mom[i], 0, sizeof(float) * sizes[i], st);
    if (e == hipSuccess) e = hipMemsetAsync(t->vel[i], 0, sizeof(float) * sizes[i], st);
  }
  if (e == hipSuccess) e = hipMalloc(&t->state, sizeof(TrainState));
  if (e == hipSuccess) {
    const TrainState s0{current_count, 0, 1.0f, 1.0f};
    e = hipMemcpy(t->state, &s0, sizeof(TrainState), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipMalloc(&t->mean64, sizeof(double) * p.in_dim);
  if (e == hipSuccess) e = hipMalloc(&t->var64, sizeof(double) * p.in_dim);
  if (e == hipSuccess && running_mean_dev)
    e = hipMemcpyAsync(t->mean64, running_mean_dev, sizeof(double) * p.in_dim, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess && running_variance_dev)
    e = hipMemcpyAsync(t->var64, running_variance_dev, sizeof(double) * p.in_dim, hipMemcpyDeviceToDevice, st);
  // workspace: see the carve-up in amp_disc_train_step
  const int64_t B = t->max_rows, M = 3 * B, Mp = up(M, 16), Bp = up(B, 16);
  const int64_t f = M * p.k1p + M * p.h1 + M * p.h2 + 2 * M + M * p.h2 + M * p.h1 +              // Xs H1 H2 logit dlogit dH2 dH1
                    (int64_t)p.h2 * Mp + 2 * (int64_t)p.h1 * Mp + (int64_t)t->kN * Mp +            // dH2T H1T dH1T XsT
                    B * p.h2 + B * p.h1 + B * t->kN + B * p.h1 + B * p.h2 +                        // a2 a1 g e1 da2
                    (int64_t)p.h1 * Bp + (int64_t)t->kN * Bp + (int64_t)p.h2 * Bp + (int64_t)p.h1 * Bp +  // a1T dgT a2T e1T
                    (int64_t)p.h1 * t->kN + p.h1 + (int64_t)p.h2 * p.h1 + p.h2 + p.h2 + 1 + 64;    // grads + loss
  t->ws_floats = f + (int64_t)16 * p.h2 * p.h1 + (int64_t)16 * p.h1 * t->kN + 64 + 2 * ((M + 3) / 4) + 16 + (int64_t)kChunks * 1024 + 1024 + 3 * (up((int64_t)kChunks * p.in_dim * 2, 8) * 2) + 16 +
                 16 * 40 + 6 * up(p.k1p, 16) + M * (t->kN - p.k1p) + 2 * (int64_t)kChunks * 1024;
  t->ws_floats += (int64_t)16 * p.h2 * p.h1 + (int64_t)16 * p.h1 * t->kN + (int64_t)3 * kChunks * 1024 + 1024 + 64;  // the side stream's slices + partials
  t->ws_floats += (int64_t)32 * p.h1 * (up(t->kN, 128) - t->kN) + B * (up(t->kN, 128) - t->kN) + 64;                  // gW1's slices / g at the padded pitch
  if (e == hipSuccess) e = hipMalloc(&t->ws, sizeof(float) * t->ws_floats);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking);
  // device-side ordering between two streams of this device only: no timing, no system-scope fence at the record (that fence
  // delayed the kernel behind a record by ~8 us in the rocprofv3 timeline of the step)
  for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&t->ev[i], hipEventDisableTiming | hipEventDisableSystemFence);
  if (e == hipSuccess && cfg->gemm_f16x3 && p.h2 % 128 == 0 && p.h1 % 128 == 0) {
    // fp16-split GEMM path: the planes of dH2, W2^T, dH2^T, H1^T (2 halves per element; batch rows padded to the k-block)
    const int64_t Mk = up(M, 32);
    e = hipMalloc(&t->dh2p, sizeof(_Float16) * 2 * Mk * p.h2);
    if (e == hipSuccess) e = hipMalloc(&t->w2tp, sizeof(_Float16) * 2 * (size_t)p.h1 * p.h2);
    if (e == hipSuccess) e = hipMalloc(&t->dh2tp, sizeof(_Float16) * 2 * Mk * p.h2);
    if (e == hipSuccess) e = hipMalloc(&t->h1tp, sizeof(_Float16) * 2 * Mk * p.h1);
    const int64_t Bk = up(B, 32), kNp = up(t->kN, 128);
    if (e == hipSuccess) e = hipMalloc(&t->dh1tp, sizeof(_Float16) * 2 * Mk * p.h1);
    if (e == hipSuccess) e = hipMalloc(&t->xstp, sizeof(_Float16) * 2 * Mk * kNp);
    if (e == hipSuccess) e = hipMemsetAsync(t->xstp, 0, sizeof(_Float16) * 2 * Mk * kNp, st);   // rows [kN, kNp) are never written again
    if (e == hipSuccess) e = hipMalloc(&t->w1tp, sizeof(_Float16) * 2 * kNp * p.h1);
    if (e == hipSuccess) e = hipMemsetAsync(t->w1tp, 0, sizeof(_Float16) * 2 * kNp * p.h1, st);     // rows [kN, kNp) stay zero
    if (e == hipSuccess) e = hipMalloc(&t->w1kp, sizeof(_Float16) * 2 * kNp * p.h1);
    if (e == hipSuccess) e = hipMalloc(&t->a1p, sizeof(_Float16) * 2 * Bk * p.h1);
    if (e == hipSuccess) e = hipMalloc(&t->a1tp, sizeof(_Float16) * 2 * Bk * p.h1);
    if (e == hipSuccess) e = hipMalloc(&t->gp, sizeof(_Float16) * 2 * Bk * kNp);
    if (e == hipSuccess) e = hipMalloc(&t->gtp, sizeof(_Float16) * 2 * Bk * kNp);
    if (e == hipSuccess) e = hipMalloc(&t->w2p, sizeof(_Float16) * 2 * (size_t)p.h1 * p.h2);
    if (e == hipSuccess) e = hipMalloc(&t->a2p, sizeof(_Float16) * 2 * Bk * p.h2);
    if (e == hipSuccess) e = hipMalloc(&t->a2tp, sizeof(_Float16) * 2 * Bk * p.h2);
    if (e == hipSuccess) e = hipMalloc(&t->e1p, sizeof(_Float16) * 2 * Bk * p.h1);
    if (e == hipSuccess) e = hipMalloc(&t->e1tp, sizeof(_Float16) * 2 * Bk * p.h1);
    if (e == hipSuccess) e = hipMalloc(&t->bound, sizeof(float) * 16);
    if (e == hipSuccess) e = hipMemsetAsync(t->bound, 0, sizeof(float) * 16, st);
    if (e == hipSuccess && f16x3_init() != AMP_OK) e = hipErrorUnknown;
  }
  if (e != hipSuccess) {
    amp_disc_trainer_destroy(t);
    return fail(AMP_ERR_HIP, "amp_disc_trainer_create: %s", hipGetErrorString(e));
  }
  if (!running_mean_dev) {
    AMP_HIP(hipMemsetAsync(t->mean64, 0, sizeof(double) * p.in_dim, st));
    // variance 1: fill through a tiny host copy
    double* ones = new double[p.in_dim];
    for (int i = 0; i < p.in_dim; ++i) ones[i] = 1.0;
    hipError_t e2 = hipMemcpy(t->var64, ones, sizeof(double) * p.in_dim, hipMemcpyHostToDevice);
    delete[] ones;
    if (e2 != hipSuccess) {
      amp_disc_trainer_destroy(t);
      return fail(AMP_ERR_HIP, "amp_disc_trainer_create: %s", hipGetErrorString(e2));
    }
  }
  *out = t;
  return AMP_OK;
}

int amp_disc_trainer_scaler(const AmpDiscTrainer* t, double* mean_out, double* var_out, double* count, amp_stream_t stream) {
  AMP_REQUIRE(t, "amp_disc_trainer_scaler: null handle");
  hipStream_t st = (hipStream_t)stream;
  if (mean_out) AMP_HIP(hipMemcpyAsync(mean_out, t->mean64, sizeof(double) * t->p.in_dim, hipMemcpyDeviceToDevice, st));
  if (var_out) AMP_HIP(hipMemcpyAsync(var_out, t->var64, sizeof(double) * t->p.in_dim, hipMemcpyDeviceToDevice, st));
  if (count) {  // the count lives on the device (a replayed graph advances it): one small blocking read-back
    TrainState s{};
    AMP_HIP(hipMemcpyAsync(&s, t->state, sizeof(TrainState), hipMemcpyDeviceToHost, st));
    AMP_HIP(hipStreamSynchronize(st));
    *count = s.count;
  }
  return AMP_OK;
}

int amp_disc_trainer_adam_state(const AmpDiscTrainer* t, float* exp_avg, float* exp_avg_sq, int64_t* step, amp_stream_t stream) {
  AMP_REQUIRE(t, "amp_disc_trainer_adam_state: null handle");
  hipStream_t st = (hipStream_t)stream;
  const DiscParams& p = t->p;
  const int64_t n[6] = {(int64_t)p.h1 * p.in_dim, p.h1, (int64_t)p.h2 * p.h1, p.h2, p.h2, 1};
  int64_t off = 0;
  for (int k = 0; k < 6; ++k) {  // the moments are stored per tensor in its logical shape (adam_multi_kernel's index i)
    if (exp_avg) AMP_HIP(hipMemcpyAsync(exp_avg + off, t->mom[k], sizeof(float) * n[k], hipMemcpyDeviceToDevice, st));
    if (exp_avg_sq) AMP_HIP(hipMemcpyAsync(exp_avg_sq + off, t->vel[k], sizeof(float) * n[k], hipMemcpyDeviceToDevice, st));
    off += n[k];
  }
  if (step) {
    TrainState s{};
    AMP_HIP(hipMemcpyAsync(&s, t->state, sizeof(TrainState), hipMemcpyDeviceToHost, st));
    AMP_HIP(hipStreamSynchronize(st));
    *step = (int64_t)s.step;
  }
  return AMP_OK;
}

int amp_disc_train_step(AmpDiscTrainer* t, const float* policy, const float* replay, const float* motion, int64_t rows,
                        int64_t row_stride, float* loss_dev, float* grads_dev, amp_stream_t stream) {
  AMP_REQUIRE(t && policy && replay && motion, "amp_disc_train_step: null argument");
  AMP_REQUIRE(rows >= 2 && rows <= t->max_rows, "amp_disc_train_step: rows per group must be in [2, %lld]", (long long)t->max_rows);
  const DiscParams p = t->p;
  AMP_REQUIRE(row_stride >= p.in_dim, "amp_disc_train_step: row_stride too small");
  hipStream_t st = (hipStream_t)stream;
  const AmpDiscTrainCfg& c = t->cfg;
  const int64_t B = rows, M = 3 * B, Mp = up(M, 16), Bp = up(B, 16);
  const int kN = t->kN, k1p = p.k1p, H1n = p.h1, H2n = p.h2;
  // ---- workspace carve-up (every region 16-float aligned) ------------------------------------------------------
  float* w = t->ws;
  auto take = [&](int64_t n) { float* r = w; w += up(n, 16); return r; };
  float* Xs = take(M * kN);      // row pitch kN (zero beyond in_dim): the TT product dH1^T Xs reads whole 64-column tiles
  float* H1 = take(M * H1n);
  float* H2 = take(M * H2n);
  float* logit = take(M);
  float* dlogit = take(M);
  float* bce_part = take(2 * ((M + 3) / 4));   // rowdot_bce_kernel's block partials (alive until the step's last kernel)
  float* dH2 = take(M * H2n);
  float* dH1 = take(M * H1n);
  float* dH2T = take((int64_t)H2n * Mp);
  float* H1T = take((int64_t)H1n * Mp);
  float* dH1T = take((int64_t)H1n * Mp);
  float* XsT = take((int64_t)kN * Mp);
  float* a2 = take(B * H2n);
  float* a1 = take(B * H1n);
  float* g = take(B * up(kN, 128));   // row pitch kN, or kN rounded up to 128 when the fp16 pipe writes it
  float* e1 = take(B * H1n);
  float* da2 = take(B * H2n);
  float* a1T = take((int64_t)H1n * Bp);
  float* dgT = take((int64_t)kN * Bp);
  float* a2T = take((int64_t)H2n * Bp);
  float* e1T = take((int64_t)H1n * Bp);
  float* gW1 = take((int64_t)H1n * kN);
  float* gb1 = take(H1n);
  float* gW2 = take((int64_t)H2n * H1n);
  float* gb2 = take(H2n);
  float* gw3 = take(H2n);
  float* gb3 = take(1);
  float* loss = take(16);  // [0] prediction, [1] gradient penalty, [2] logit reg, [3] weight decay
  float* split = take((int64_t)32 * H2n * H1n);                  // split-K partial products of gW2: <= 16 slices per product, two products
  float* split1 = take((int64_t)32 * H1n * up(kN, 128));         // ... of gW1 (both weights' slices are alive at once: deferred sums; row
                                                                  //     pitch kN, or kN rounded up to 128 when the fp16 pipe writes them)
  float* part = take((int64_t)3 * kChunks * 1024 + 1024);      // column-sum / scalar partials (three planes: dh2_colsum_kernel)
  float* part_side = take((int64_t)3 * kChunks * 1024 + 1024); // ... of the gradient-penalty chain when it runs on the side stream
  const int64_t dpart_stride = up((int64_t)kChunks * p.in_dim * 2, 8);                 // doubles per batch
  double* dpart = reinterpret_cast<double*>(take(3 * dpart_stride * 2 + 16));
  const int vec_stride = (int)up(k1p, 16);
  float* mean32w = take(3 * vec_stride);  // the running statistics after each batch, as the scaling pass reads them
  float* den32w = take(3 * vec_stride);
  AMP_REQUIRE(w - t->ws <= t->ws_floats, "amp_disc_train_step: internal workspace overflow");
  AMP_REQUIRE(H1n <= 1024 && H2n <= 1024, "amp_disc_train_step: hidden sizes above 1024 are not supported");
  auto colsum = [&](const float* A, int64_t rows_, int cols_, int64_t lda_, const float* rowscale, const float* mask, int64_t ldm,
                    float* out, int accumulate) {
    colsum_part_kernel<<<dim3((cols_ + 63) / 64, kChunks), kBlock, 0, st>>>(A, rows_, cols_, lda_, rowscale, mask, ldm, part);
    colsum_final_kernel<<<(cols_ + kBlock - 1) / kBlock, kBlock, 0, st>>>(part, cols_, out, accumulate);
  };
  auto sumsq = [&](hipStream_t s_, float* part_, float* x, int64_t rows_, int cols_, int64_t ld_, float coef, int in_place, float scale,
                   int slot, int accumulate) {
    const int nb = (int)std::min<int64_t>(rows_, 1024);
    sumsq_part_kernel<<<nb, kBlock, 0, s_>>>(x, rows_, cols_, ld_, coef, in_place, part_);
    scalar_final_kernel<<<1, 256, 0, s_>>>(part_, nb, scale, loss, slot, accumulate);
  };
  int rc;
  // layer 1's bias gradient = colsum(dH1) comes out of the TT product dH1^T Xs when Xs carries a column of ones in its padding
  // (column in_dim; needs a spare column and the TT route: the fp16-split option keeps the separate column sum)
  const int ones_col = p.in_dim < kN ? p.in_dim : -1;
  // cfg.gemm_f16x3: the two large products of the prediction loss's backward (dH1, gW2: 2 x 12.9 of the backward's 30.6 GFLOP on the
  // caller's stream) run at fp32 accuracy on the fp16 matrix pipe; everything else -- the two FORWARD GEMMs, the products of output
  // width kN = 192, the gradient-penalty chain on the side stream -- stays on the fp32 pipe.
  // The forward decides the ReLU masks: a pre-activation within rounding of zero flips its mask with any change of the
  // summation (measured: one of 786 432 H2 entries between the two engines at 3 x 512 rows), and one flipped unit moves its
  // column of the bias / weight gradients by ~1 / sqrt(rows) of the column sum -- 6e-3 here, two orders above the parity bar
  // although both forwards are 1e-6 from fp64.  The backward products are linear in their operands: no such cliff.
  auto nt_on = [&](hipStream_t s_, const float* A, int64_t lda, int64_t Mr, const float* W, int Kp, int N, float* C, int64_t ldc,
                   const float* mask, int64_t ldmask, int accumulate, float* split_ws = nullptr) -> int {
    return gemm_nt(s_, A, lda, Mr, W, Kp, N, C, ldc, mask, ldmask, accumulate, split_ws);
  };
  auto nt = [&](const float* A, int64_t lda, int64_t Mr, const float* W, int Kp, int N, float* C, int64_t ldc, const float* mask,
                int64_t ldmask, int accumulate, float* split_ws = nullptr) -> int {
    return nt_on(st, A, lda, Mr, W, Kp, N, C, ldc, mask, ldmask, accumulate, split_ws);
  };
  // ---- fork: the gradient-penalty chain on the trainer's side stream when all four weight-gradient products take the TT kernel
  const bool pair = c.grad_penalty_scale != 0.0f;
  int sl_w2[2] = {0, 0}, sl_w1[2] = {0, 0};   // k-slices of (prediction, penalty) product of gW2 / gW1
  bool fork = pair && fork_ok() && t->side != nullptr;
  if (fork) {
    TtPlan q;
    fork = tt_plan(H2n, H1n, M, H2n, H1n, H1n, true, &q) == AMP_OK;
    sl_w2[0] = q.slices;
    fork = fork && tt_plan(H1n, kN, M, H1n, kN, kN, true, &q) == AMP_OK;
    sl_w1[0] = q.slices;
    fork = fork && tt_plan(H1n, kN, B, H1n, kN, kN, true, &q) == AMP_OK;
    sl_w1[1] = q.slices;
    fork = fork && tt_plan(H2n, H1n, B, H2n, H1n, H1n, true, &q) == AMP_OK;
    sl_w2[1] = q.slices;
  }
  // fp16-split products (forked step only: their planes are written on the side stream): dH1 [M, h1] over K = h2, gW2 [h2, h1] over
  // K = the batch rows padded to the k-block (zero rows), its k-slices summed by Adam like the TT product's
  const int64_t Mk = up(M, 32);
  const bool f16 = fork && t->dh2p != nullptr && M >= 128 && H2n % 32 == 0;
  // ... and of the penalty chain (B motion rows): a1 [B, h1] over K = h2, gW2's second product [h2, h1] over K = B (padded), da2
  // [B, h2] over K = h1
  const int64_t Bk = up(B, 32);
  const bool f16p = f16 && B >= 128 && H1n % 32 == 0;
  F16Plan plan_dh1{}, plan_gw2{}, plan_a1{}, plan_gw2p{}, plan_da2{}, plan_gw1{};
  const int kNp = (int)up(kN, 128);
  const int64_t gN = f16 ? kNp : kN;   // row pitch of gW1's k-slices
  if (f16) {
    plan_dh1 = f16_planes_plan(M, H1n, H2n / 32, 1);
    plan_gw2 = f16_planes_plan(H2n, H1n, (int)(Mk / 32), 16);
    plan_gw1 = f16_planes_plan(H1n, kNp, (int)(Mk / 32), 16);
    sl_w2[0] = plan_gw2.slices;
    sl_w1[0] = plan_gw1.slices;
  }
  // ... and, for a WIDE input only (K D = 830: kN = 832), its other three: g = a1 W1, gW1's second product a1^T g, e1 = g W1^T.  At
  // kN = 192 they are 1.6 GFLOP each: the fp32 kernels take 25-35 us beside the other stream, the fp16 ones the same plus their operand
  // passes, and more LDS-DMA workgroups on the side stream slow the caller's fp16 products down (K D = 166: 0.568 -> 0.607 ms per
  // step, same box; K D = 830: 0.989 -> 0.896, profiles/r05_train_step.md)
  const bool f16g = f16p && kN >= 512;
  F16Plan plan_g{}, plan_e1{}, plan_gw1p{};
  const int64_t gP = f16g ? kNp : kN;   // row pitch of g
  if (f16g) {
    plan_g = f16_planes_plan(B, kNp, H1n / 32, 1);
    plan_e1 = f16_planes_plan(B, H1n, kNp / 32, 1);
    plan_gw1p = f16_planes_plan(H1n, kNp, (int)(Bk / 32), 16);
    sl_w1[1] = plan_gw1p.slices;
  }
  if (f16p) {
    plan_a1 = f16_planes_plan(B, H1n, H2n / 32, 1);
    plan_gw2p = f16_planes_plan(H2n, H1n, (int)(Bk / 32), 16);
    plan_da2 = f16_planes_plan(B, H2n, H1n / 32, 1);
    sl_w2[1] = plan_gw2p.slices;
  }
  hipStream_t side = fork ? t->side : st;
  // whatever path leaves this function, the side stream has joined the caller's stream (an error return between fork and join
  // would otherwise leave a stream capture of the step with unjoined work)
  struct ForkGuard {
    hipStream_t st, side; hipEvent_t ev; bool armed;
    ~ForkGuard() {
      if (!armed) return;
      (void)hipEventRecord(ev, side);
      (void)hipStreamWaitEvent(st, ev, 0);
    }
  } fork_guard{st, side, t->ev[3], fork};
  if (fork) {
    // the side stream starts with the two W^T copies (they only need last step's Adam), under the scaler passes
    AMP_HIP(hipEventRecord(t->ev[0], st));
    AMP_HIP(hipStreamWaitEvent(side, t->ev[0], 0));
    transpose(side, p.w2, H2n, H1n, H1n, t->w2t, H2n, H1n);               // W2^T [h1, h2]
    transpose(side, p.w1p, H1n, k1p, k1p, t->w1t, H1n, kN);               // W1^T [kN, h1] (zero rows >= k1p)
    if (f16) {
      // the planes of W2^T (the W operand of dH1) with the scale of max|W2|: 2 MB, under the scaler passes
      // (abs-max kernels: one atomic per workgroup on one address -- ~50 ns each in the L2 --, so few, long workgroups)
      AMP_HIP(hipMemsetAsync(t->bound + 1, 0, 12 * sizeof(float), side));
      amax_flat(side, p.w2, (int64_t)H2n * H1n, t->bound + 1);
      amax_flat(side, p.w3, H2n, t->bound + 3);
      rowsum_bound_kernel<<<(unsigned)((H1n + 3) / 4), kBlock, 0, side>>>(p.w1p, H1n, k1p, k1p, p.b1, t->bound + 8);
      split_transpose_blocks_kernel<<<dim3(H1n / 64, H2n / 32), kBlock, 0, side>>>(p.w2, H2n, H1n, H1n, t->bound + 1, t->w2tp, H2n);
      if (f16p) {
        split_rows_blocks_kernel<<<blocks((int64_t)H2n * (H1n / 4)), kBlock, 0, side>>>(p.w2, H2n, H1n, H1n, t->bound + 1, t->w2p, H1n, 1);
      }
      if (f16g) {
        // W1 in both orientations (t->w1t = W1^T [kN, h1] was written just above), one scale: max|W1|;  |a1| <= h2 max|w3| max|W2|
        amax_flat(side, p.w1p, (int64_t)H1n * k1p, t->bound + 10);
        split_rows_blocks_kernel<<<blocks((int64_t)kN * (H1n / 4)), kBlock, 0, side>>>(t->w1t, kN, H1n, H1n, t->bound + 10, t->w1tp, H1n, 1);
        split_rows_blocks_kernel<<<blocks((int64_t)H1n * (kNp / 4)), kBlock, 0, side>>>(p.w1p, H1n, k1p, k1p, t->bound + 10, t->w1kp, kNp, 1);
        bound_affine_kernel<<<1, 1, 0, side>>>(t->bound + 11, t->bound + 3, t->bound + 1, nullptr, (float)H2n);
      }
    }
    AMP_HIP(hipEventRecord(t->ev[1], side));
  }
  // ---- 1. scaler (train=True): update the running statistics with each batch, then scale it --------------------
  const float* mean32 = nullptr;
  const float* den32 = nullptr;
  float clip = 0.0f;
  if (c.use_scaler && !c.update_scaler) {
    // frozen statistics: the fp32 vectors already live in the discriminator handle
    rc = amp_disc_set_scaler(t->disc, t->mean64, t->var64, c.scaler_epsilon, c.scaler_clip, stream);
    if (rc != AMP_OK) return rc;
    AmpDiscInputLayout lay;
    amp_disc_input_layout(t->disc, &lay);
    mean32 = lay.mean_dev; den32 = lay.den_dev; clip = lay.clip;
  }
  {
    const Batch3 b3{{policy, replay, motion}};
    int vstride = 0;  // frozen / no statistics: one vector pair for the three batches
    if (c.update_scaler) {
      // each batch updates the running statistics, then is scaled with them (skrl order): partial sums of the three batches,
      // one in-order merge that also writes the fp32 vectors of every intermediate state, one scaling pass
      scaler_part3_kernel<<<dim3((p.in_dim + 63) / 64, kChunks, 3), kBlock, 0, st>>>(b3, B, p.in_dim, row_stride, dpart, dpart_stride);
      scaler_merge3_kernel<<<(k1p + kBlock - 1) / kBlock, kBlock, 0, st>>>(dpart, dpart_stride, B, p.in_dim, t->mean64, t->var64, t->state,
                                                                          k1p, c.scaler_epsilon, c.use_scaler ? mean32w : nullptr,
                                                                          den32w, vec_stride);
      if (c.use_scaler) { mean32 = mean32w; den32 = den32w; clip = c.scaler_clip; vstride = vec_stride; }
    }
    AMP_REQUIRE(B * kN < (int64_t)1 << 31, "amp_disc_train_step: batch too large for the scaling pass's 32-bit indices");
    scale_rows3_kernel<<<dim3(blocks(B * kN), 3), kBlock, 0, st>>>(b3, row_stride, B, p.in_dim, kN, mean32, den32, vstride, clip, Xs,
                                                                   ones_col);
  }
  const bool scaler_to_handle = c.use_scaler && c.update_scaler && !c.defer_refresh;
  if (scaler_to_handle && !fork) {
    // the discriminator handle serves inference with the statistics after the third batch
    rc = amp_disc_set_scaler(t->disc, t->mean64, t->var64, c.scaler_epsilon, c.scaler_clip, stream);
    if (rc != AMP_OK) return rc;
  }
  rc = launch_status("scale_rows3_kernel");
  if (rc != AMP_OK) return rc;

  // ---- 2. forward, keeping H1 / H2 -----------------------------------------------------------------------------
  rc = gemm_fwd(st, Xs, kN, M, p.w1p, k1p, H1n, p.b1, H1);
  if (rc != AMP_OK) return rc;
  if (f16) {
    // H1 is complete: its abs-max and the planes of H1^T (the W operand of gW2; 50 MB in, 50 MB out) on the side stream, under
    // layer 2's fp32 forward
    AMP_HIP(hipEventRecord(t->ev[4], st));
    AMP_HIP(hipStreamWaitEvent(side, t->ev[4], 0));
    // |H1| <= max_j sum_k |W1[j][k]| x max|Xs| + max|b1|: an a-priori bound (a 50-MB abs-max pass over H1 under the forward
    // of layer 2 cost that GEMM ~12 us); max|Xs| is needed for Xs^T's planes anyway
    amax_flat(side, Xs, M * kN, t->bound + 6);
    bound_affine_kernel<<<1, 1, 0, side>>>(t->bound + 2, t->bound + 8, t->bound + 6, t->bound + 9);
    split_transpose_blocks_kernel<<<dim3(H1n / 64, (unsigned)(Mk / 32)), kBlock, 0, side>>>(H1, M, H1n, H1n, t->bound + 2, t->h1tp, Mk);
    // ... and of Xs^T (the W operand of gW1; the column of ones that yields gb1 included)
    split_transpose_blocks_kernel<<<dim3(kN / 64, (unsigned)(Mk / 32)), kBlock, 0, side>>>(Xs, M, kN, kN, t->bound + 6, t->xstp, Mk);
    AMP_HIP(hipEventRecord(t->ev[5], side));
  }
  rc = gemm_fwd(st, H1, H1n, M, p.w2, H1n, H2n, p.b2, H2);
  if (rc != AMP_OK) return rc;
  const unsigned n_bce = (unsigned)((M + 3) / 4);
  if (fork) AMP_HIP(hipEventRecord(t->ev[2], st));   // H1 / H2 are complete: the penalty chain may start
  rowdot_bce_kernel<<<n_bce, kBlock, 0, st>>>(H2, M, H2n, H2n, p.w3, p.b3, 2 * B, c.loss_scale, logit, dlogit, bce_part);

  // ---- 4. gradient penalty on the motion rows (on the side stream when forked; else in line after section 3) ----
  int def_w1 = 0, def_w2 = 0;
  auto tt = [&](const float* A, int64_t lda, int Mo, const float* W, int64_t ldw, int No, int64_t Kr, float* C, int64_t ldc, int acc,
                float* sp, int* defer) -> int {
    return gemm_tt(st, A, lda, Mo, W, ldw, No, Kr, C, ldc, acc, sp, pair ? defer : nullptr);
  };
  const int64_t n_w2 = (int64_t)H2n * H1n, n_w1 = (int64_t)H1n * gN;
  auto penalty = [&]() -> int {
    hipStream_t s_ = side;
    float* part_ = fork ? part_side : part;
    const float* H1m = H1 + 2 * B * H1n;
    const float* H2m = H2 + 2 * B * H2n;
    int r;
    a2_kernel<<<blocks(B * H2n), kBlock, 0, s_>>>(p.w3, H2m, B, H2n, a2);
    if (f16p) {
      // a2 = w3 * (H2 > 0): |a2| <= max|w3| exactly; its planes in both orientations, then a1 on the fp16 pipe against W2^T's planes
      split_transpose_blocks_kernel<<<dim3(H2n / 64, (unsigned)(Bk / 32)), kBlock, 0, s_>>>(a2, B, H2n, H2n, t->bound + 3, t->a2tp, Bk, t->a2p);
      r = gemm_f16_planes(s_, t->a2p, B, t->w2tp, H1n, H2n, t->bound + 3, t->bound + 1, a1, H1n, H1m, H1n, plan_a1, 0);
    } else {
      r = nt_on(s_, a2, H2n, B, t->w2t, H2n, H1n, a1, H1n, H1m, H1n, 0);        // a1 = (a2 W2) * m1
    }
    if (r != AMP_OK) return r;
    if (f16g) {
      // a1's planes in both orientations (bound: h2 max|w3| max|W2|), then g = a1 W1 on the fp16 pipe (N padded to kNp: zero rows of W1^T)
      split_transpose_blocks_kernel<<<dim3(H1n / 64, (unsigned)(Bk / 32)), kBlock, 0, s_>>>(a1, B, H1n, H1n, t->bound + 11, t->a1tp, Bk, t->a1p);
      r = gemm_f16_planes(s_, t->a1p, B, t->w1tp, kNp, H1n, t->bound + 11, t->bound + 10, g, gP, nullptr, 0, plan_g, 0);
    } else {
      r = nt_on(s_, a1, H1n, B, t->w1t, H1n, kN, g, kN, nullptr, 0, 0);         // g = a1 W1      [B, kN]
    }
    if (r != AMP_OK) return r;
    // loss[1] = gp_scale * mean_rows |g|^2 ;  g <- dL/dg = (2 gp_scale loss_scale / B) g
    sumsq(s_, part_, g, B, p.in_dim, gP, 2.0f * c.grad_penalty_scale * c.loss_scale / (float)B, 1, c.grad_penalty_scale / (float)B, 1, 0);
    if (f16g) {
      // g's planes in both orientations (abs-max of 4 MB), gW1's penalty slices = a1^T g on the fp16 pipe
      amax_flat(s_, g, B * gP, t->bound + 12);
      split_transpose_blocks_kernel<<<dim3(kNp / 64, (unsigned)(Bk / 32)), kBlock, 0, s_>>>(g, B, kNp, gP, t->bound + 12, t->gtp, Bk, t->gp);
      r = gemm_f16_planes(s_, t->a1tp, H1n, t->gtp, kNp, (int)Bk, t->bound + 11, t->bound + 12, split1 + sl_w1[0] * n_w1, gN, nullptr, 0,
                          plan_gw1p, n_w1);
    } else if (fork) {
      int n = 0;
      r = gemm_tt(s_, a1, H1n, H1n, g, kN, kN, B, nullptr, gN, 0, split1 + sl_w1[0] * n_w1, nullptr, &n);   // gW1's penalty slices
    } else {
      r = tt(a1, H1n, H1n, g, kN, kN, B, gW1, kN, 1, split1, &def_w1);        // gW1 += a1^T dg
      if (r == kShapeNotSupported) {
        transpose(st, a1, B, H1n, H1n, a1T, Bp, H1n);
        transpose(st, g, B, kN, kN, dgT, Bp, kN);
        r = nt(a1T, Bp, H1n, dgT, (int)Bp, kN, gW1, kN, nullptr, 0, 1, split);
      }
    }
    if (r != AMP_OK) return r;
    if (f16g) r = gemm_f16_planes(s_, t->gp, B, t->w1kp, H1n, kNp, t->bound + 12, t->bound + 10, e1, H1n, H1m, H1n, plan_e1, 0);
    else r = nt_on(s_, g, kN, B, p.w1p, k1p, H1n, e1, H1n, H1m, H1n, 0);      // e1 = (dg W1^T) * m1   (K = k1p <= kN)
    if (r != AMP_OK) return r;
    if (f16p) {
      // e1's abs-max (16 MB) and planes in both orientations; gW2's penalty slices = a2^T e1 and da2 = e1 W2^T on the fp16 pipe
      amax_flat(s_, e1, B * H1n, t->bound + 4);   // (slot zeroed at the head of the side stream's work)
      split_transpose_blocks_kernel<<<dim3(H1n / 64, (unsigned)(Bk / 32)), kBlock, 0, s_>>>(e1, B, H1n, H1n, t->bound + 4, t->e1tp, Bk, t->e1p);
      r = gemm_f16_planes(s_, t->a2tp, H2n, t->e1tp, H1n, (int)Bk, t->bound + 3, t->bound + 4, split + sl_w2[0] * n_w2, H1n, nullptr, 0,
                          plan_gw2p, n_w2);
    } else if (fork) {
      int n = 0;
      r = gemm_tt(s_, a2, H2n, H2n, e1, H1n, H1n, B, nullptr, H1n, 0, split + sl_w2[0] * n_w2, nullptr, &n);  // gW2's penalty slices
    } else {
      r = tt(a2, H2n, H2n, e1, H1n, H1n, B, gW2, H1n, 1, split, &def_w2);    // gW2 += a2^T e1
      if (r == kShapeNotSupported) {
        transpose(st, a2, B, H2n, H2n, a2T, Bp, H2n);
        transpose(st, e1, B, H1n, H1n, e1T, Bp, H1n);
        r = nt(a2T, Bp, H2n, e1T, (int)Bp, H1n, gW2, H1n, nullptr, 0, 1, split);
      }
    }
    if (r != AMP_OK) return r;
    if (f16p) r = gemm_f16_planes(s_, t->e1p, B, t->w2p, H2n, H1n, t->bound + 4, t->bound + 1, da2, H2n, nullptr, 0, plan_da2, 0);
    else r = nt_on(s_, e1, H1n, B, p.w2, H1n, H2n, da2, H2n, nullptr, 0, 0);       // da2 = e1 W2^T
    if (r != AMP_OK) return r;
    // gw3 += colsum(m2 * da2): the partial sums here, the accumulation into gw3 after the join when forked
    colsum_part_kernel<<<dim3((H2n + 63) / 64, kChunks), kBlock, 0, s_>>>(da2, B, H2n, H2n, nullptr, H2m, H2n, part_);
    if (!fork) colsum_final_kernel<<<(H2n + kBlock - 1) / kBlock, kBlock, 0, s_>>>(part_, H2n, gw3, 1);
    return launch_status("colsum_part_kernel");
  };
  if (fork) {
    AMP_HIP(hipStreamWaitEvent(side, t->ev[2], 0));
    // advance the device-side state here (the scaler's merge, which reads the count, is behind ev[2]; Adam, which reads the bias
    // corrections, is behind the join): off the step's critical path
    train_state_kernel<<<1, 1, 0, side>>>(t->state, c.update_scaler ? 3.0 * (double)B : 0.0, (double)c.beta1, (double)c.beta2);
    rc = penalty();
    if (rc != AMP_OK) return rc;
    if (scaler_to_handle) {   // the handle's copy of the statistics (read by inference only, which is behind the join); at the END of
                              // the side stream's work: in front of the chain its two small launches delayed every GEMM behind them
      rc = amp_disc_set_scaler(t->disc, t->mean64, t->var64, c.scaler_epsilon, c.scaler_clip, (amp_stream_t)side);
      if (rc != AMP_OK) return rc;
    }
    AMP_HIP(hipEventRecord(t->ev[3], side));
  }

  // ---- 3. backward of the prediction loss ----------------------------------------------------------------------
  // dH2 = dlogit (x) w3 * (H2 > 0), gw3 = H2^T dlogit, gb2 = colsum(dH2), gb3 = sum(dlogit): one pass over H2
  if (f16) {
    AMP_HIP(hipStreamWaitEvent(st, t->ev[1], 0));   // max|W2| (bound[1]) is in place
    // |dH2| <= max|dlogit| max|w3|;  |dH1| <= h2 max|dH2| max|W2|
    bound_product_kernel<<<1, 1024, 0, st>>>(dlogit, M, p.w3, H2n, t->bound, t->bound + 1, (float)H2n, t->bound + 5);
  }
  dh2_colsum_kernel<<<dim3((H2n + 63) / 64, kChunks), kBlock, 0, st>>>(dlogit, p.w3, H2, M, H2n, dH2, part, f16 ? t->dh2p : nullptr, t->bound);
  dh2_colsum_final_kernel<<<(H2n + kBlock - 1) / kBlock, kBlock, 0, st>>>(part, H2n, gw3, gb2, gb3);
  if (fork) {
    AMP_HIP(hipStreamWaitEvent(st, t->ev[1], 0));
  } else {
    transpose(st, p.w2, H2n, H1n, H1n, t->w2t, H2n, H1n);               // W2^T [h1, h2]
    transpose(st, p.w1p, H1n, k1p, k1p, t->w1t, H1n, kN);               // W1^T [kN, h1] (zero rows >= k1p)
  }
  if (f16) {
    // dH2's planes in both orientations (one scale), then dH1 = (dH2 W2) * (H1 > 0) on the fp16 pipe
    // (dH2's row planes came out of dh2_colsum_kernel; the transposed ones -- gW2's A operand -- behind dH1's launch)
    rc = gemm_f16_planes(st, t->dh2p, M, t->w2tp, H1n, H2n, t->bound, t->bound + 1, dH1, H1n, H1, H1n, plan_dh1, 0);
  } else {
    rc = nt(dH2, H2n, M, t->w2t, H2n, H1n, dH1, H1n, H1, H1n, 0);   // dH1 = (dH2 W2) * (H1 > 0)
  }
  if (rc != AMP_OK) return rc;
  if (ones_col < 0) colsum(dH1, M, H1n, H1n, nullptr, nullptr, 0, gb1, 0);
  // the weight gradients reduce over the batch: the TT kernel takes both operands as the kernels above left them (rows = batch)
  // (the fp16-split option keeps its transposed-copy route: its kernel is k-contiguous by construction)
  // each weight's two products (prediction loss, gradient penalty) share one slice sum: in line the first defers and the second
  // accumulates into its slices; forked, the two write slice regions of their own and one sum follows the join
  if (fork) {
    int n = 0;
    if (f16) {
      split_transpose_blocks_kernel<<<dim3(H2n / 64, (unsigned)(Mk / 32)), kBlock, 0, st>>>(dH2, M, H2n, H2n, t->bound, t->dh2tp, Mk);
      AMP_HIP(hipStreamWaitEvent(st, t->ev[5], 0));   // H1^T's planes (written under the forward of layer 2)
      rc = gemm_f16_planes(st, t->dh2tp, H2n, t->h1tp, H1n, (int)Mk, t->bound, t->bound + 2, split, H1n, nullptr, 0, plan_gw2, n_w2);
    } else {
      rc = gemm_tt(st, dH2, H2n, H2n, H1, H1n, H1n, M, nullptr, H1n, 0, split, nullptr, &n);      // gW2 = dH2^T H1
    }
    if (rc != AMP_OK) return rc;
    if (f16) {
      split_transpose_blocks_kernel<<<dim3(H1n / 64, (unsigned)(Mk / 32)), kBlock, 0, st>>>(dH1, M, H1n, H1n, t->bound + 5, t->dh1tp, Mk);
      rc = gemm_f16_planes(st, t->dh1tp, H1n, t->xstp, kNp, (int)Mk, t->bound + 5, t->bound + 6, split1, gN, nullptr, 0, plan_gw1, n_w1);
    } else {
      rc = gemm_tt(st, dH1, H1n, H1n, Xs, kN, kN, M, nullptr, kN, 0, split1, nullptr, &n);        // gW1 = dH1^T Xs (+ gb1 in column ones_col)
    }
    if (rc != AMP_OK) return rc;
    AMP_HIP(hipStreamWaitEvent(st, t->ev[3], 0));   // join; Adam sums the slices and the penalty's w3 column partials itself
    fork_guard.armed = false;
  } else {
    rc = tt(dH2, H2n, H2n, H1, H1n, H1n, M, gW2, H1n, 0, split, &def_w2);       // gW2 = dH2^T H1
    if (rc == kShapeNotSupported) {
      transpose(st, dH2, M, H2n, H2n, dH2T, Mp, H2n);
      transpose(st, H1, M, H1n, H1n, H1T, Mp, H1n);
      rc = nt(dH2T, Mp, H2n, H1T, (int)Mp, H1n, gW2, H1n, nullptr, 0, 0, split);
    }
    if (rc != AMP_OK) return rc;
    rc = tt(dH1, H1n, H1n, Xs, kN, kN, M, gW1, kN, 0, split1, &def_w1);          // gW1 = dH1^T Xs (+ gb1 in column ones_col)
    if (rc == kShapeNotSupported) {
      transpose(st, dH1, M, H1n, H1n, dH1T, Mp, H1n);
      transpose(st, Xs, M, k1p, kN, XsT, Mp, kN);
      rc = nt(dH1T, Mp, H1n, XsT, (int)Mp, kN, gW1, kN, nullptr, 0, 0, split);
    }
    if (rc != AMP_OK) return rc;
    if (pair) {
      rc = penalty();
      if (rc != AMP_OK) return rc;
    } else {
      AMP_HIP(hipMemsetAsync(loss + 1, 0, sizeof(float), st));
    }
  }

  // ---- 5 + 6. Adam; the regularisers' values for the report come out of the same pass over the parameters (their gradients
  //              are folded into the update) ------------------------------------------------------------------------
  // advance the device-side state: scaler count += the batches merged above, Adam step += 1, bias corrections of this step
  if (!fork) train_state_kernel<<<1, 1, 0, st>>>(t->state, c.update_scaler ? 3.0 * (double)B : 0.0, (double)c.beta1, (double)c.beta2);
  const float wd2 = 2.0f * c.loss_scale * c.weight_decay_scale, lr2 = 2.0f * c.loss_scale * c.logit_reg_scale;
  const float lr = c.apply_update ? c.learning_rate : 0.0f;
  {
    AdamSegs a{};
    float* ps[6] = {p.w1p, p.b1, p.w2, p.b2, p.w3, p.b3};
    // b1's gradient: gb1, or column ones_col of gW1 (one element per row of gW1)
    const bool col = ones_col >= 0;
    // forked: W1 / W2 (and b1 in W1's column of ones) are still k-slices in `split1` / `split`
    const float* gw1_src = fork ? split1 : gW1;
    const float* gs[6] = {gw1_src, col ? gw1_src + ones_col : gb1, fork ? split : gW2, gb2, gw3, gb3};
    const int nsl[6] = {fork ? sl_w1[0] + sl_w1[1] : 1, (fork && col) ? sl_w1[0] + sl_w1[1] : 1, fork ? sl_w2[0] + sl_w2[1] : 1, 1, 1, 1};
    const int64_t sst[6] = {n_w1, n_w1, n_w2, 0, 0, 0};
    a.colpart = fork ? part_side : nullptr;
    a.colpart_chunks = kChunks;
    const int64_t ldp[6] = {k1p, col ? 1 : H1n, H1n, H2n, H2n, 1}, ldg[6] = {fork ? gN : kN, col ? (fork ? gN : kN) : H1n, H1n, H2n, H2n, 1};
    const int64_t rows_[6] = {H1n, col ? H1n : 1, H2n, 1, 1, 1};
    const int cols_[6] = {p.in_dim, col ? 1 : H1n, H1n, H2n, H2n, 1};
    const float reg[6] = {wd2, 0.0f, wd2, 0.0f, wd2 + lr2, 0.0f};
    a.start[0] = 0;
    for (int k = 0; k < 6; ++k) {
      a.p[k] = ps[k]; a.g[k] = gs[k]; a.m[k] = t->mom[k]; a.v[k] = t->vel[k];
      a.ld_p[k] = ldp[k]; a.ld_g[k] = ldg[k]; a.cols[k] = cols_[k]; a.reg2[k] = reg[k];
      a.slices[k] = nsl[k]; a.slice_stride[k] = sst[k];
      a.start[k + 1] = a.start[k] + rows_[k] * cols_[k];
    }
    const unsigned nb = blocks(a.start[6]);
    AMP_REQUIRE((int64_t)3 * nb <= (int64_t)3 * kChunks * 1024 + 1024, "amp_disc_train_step: too many Adam blocks for the partials buffer");
    adam_multi_kernel<<<nb, kBlock, 0, st>>>(a, lr, c.beta1, c.beta2, c.adam_epsilon, t->state, grads_dev, part);
    reg_final_kernel<<<1, kFinalBlock, 0, st>>>(part, (int)nb, c.logit_reg_scale, c.weight_decay_scale, c.loss_scale, bce_part, (int)n_bce,
                                        (float)(2 * B), (float)B, loss, loss_dev);
  }
  rc = launch_status("adam_multi_kernel");
  if (rc != AMP_OK) return rc;
  return (c.apply_update && !c.defer_refresh) ? disc_refresh_derived(t->disc, st) : AMP_OK;
}

int amp_disc_train_tt_plan(int32_t M, int32_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int32_t split, int32_t* bm,
                           int32_t* bn, int32_t* slices) {
  AMP_REQUIRE(bm && bn && slices, "amp_disc_train_tt_plan: null output");
  TtPlan p;
  if (tt_plan(M, N, K, lda, ldw, ldc, split != 0, &p) != AMP_OK)
    return amp::fail(AMP_ERR_INVALID, "amp_disc_train_tt_plan: shape not admitted (M = %d, N = %d, K = %lld, pitches %lld / %lld / %lld)", M, N,
                     (long long)K, (long long)lda, (long long)ldw, (long long)ldc);
  *bm = p.bm; *bn = p.bn; *slices = p.slices;
  return AMP_OK;
}

int amp_disc_trainer_refresh(AmpDiscTrainer* t, amp_stream_t stream) {
  AMP_REQUIRE(t, "amp_disc_trainer_refresh: null handle");
  const AmpDiscTrainCfg& c = t->cfg;
  if (c.use_scaler && c.update_scaler) {
    const int rc = amp_disc_set_scaler(t->disc, t->mean64, t->var64, c.scaler_epsilon, c.scaler_clip, stream);
    if (rc != AMP_OK) return rc;
  }
  return c.apply_update ? disc_refresh_derived(t->disc, (hipStream_t)stream) : AMP_OK;
}

}  // extern "C"
