// Device ring buffers of AMP observation rows: the agent-side stores of skrl's AMP [third-party, absent: parity
// unpinned] -- `reply_buffer` (1 M rows) and `motion_dataset` (200 k rows), both skrl `RandomMemory` objects
// (agents/skrl_g1_walk_amp_cfg.yaml:44-58) that the discriminator update samples `discriminator_batch_size` rows from
// (SURVEY.md section 8f rank 1).  Semantics restated from skrl: `add_samples` writes a batch at the write head and
// wraps around; `sample` draws indices uniformly WITH replacement over the rows written so far.
//
//   append  n rows (any row stride) -> storage[(head + i) % capacity]; head / size live on the host (the caller
//           serialises appends on one stream, like every other engine call)
//   sample  row i = storage[floor(u_g * size)], g = first_row + i, u_g = word 0 of Philox4x32-10(counter = (g, draw), key = seed):
//           reproducible, order-independent, no host RNG; indices optionally returned (bit-exact vs oracle/rng.py).
//           `first_row` is the GLOBAL position of this call's first row in the minibatch the rows belong to: rank w of a
//           multi-rank discriminator update draws its rows [w * n, (w + 1) * n) of the minibatch, so the variates of a
//           minibatch do not depend on how many ranks share it (humanoid_amp_amd/distributed.py)
#include "amp_common.hpp"

struct AmpRing {
  int64_t capacity;
  int32_t dim;
  int64_t head, size;
  float* rows;  // [capacity, dim]
};

namespace amp {

__global__ __launch_bounds__(kBlock) void ring_append_kernel(const float* __restrict__ src, int64_t n, int64_t stride, int dim,
                                                             float* __restrict__ rows, int64_t head, int64_t capacity) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n * dim) return;
  const int64_t i = e / dim;
  const int c = (int)(e - i * dim);
  int64_t r = head + i;
  r -= r >= capacity ? capacity : 0;
  rows[r * dim + c] = src[i * stride + c];
}

__global__ __launch_bounds__(kBlock) void ring_sample_kernel(const float* __restrict__ rows, int64_t size, int dim, uint64_t seed,
                                                             uint64_t draw, int64_t first, int64_t n, float* __restrict__ out,
                                                             int64_t out_stride, int64_t* __restrict__ idx_out) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n * dim) return;
  const int64_t i = e / dim;
  const int c = (int)(e - i * dim);
  const uint64_t g = (uint64_t)(first + i);
  uint32_t r[4];
  philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)draw, (uint32_t)(draw >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), r);
  // size < 2^32: floor(u * size) with u = r0 / 2^32 (bias <= size / 2^32, as torch.randint's modulo has)
  const int64_t row = (int64_t)(((uint64_t)r[0] * (uint64_t)size) >> 32);
  out[i * out_stride + c] = rows[row * dim + c];
  if (idx_out && c == 0) idx_out[i] = row;
}

// Pseudo-random PERMUTATION of [0, n) evaluated point-wise (no sort, no n-element index array): a 6-round balanced Feistel network
// over 2 * hb bits (2^(2 hb) >= n, < 4 n) keyed by (seed, epoch), cycle-walked until the image falls inside [0, n) -- a bijection of
// the 2^(2 hb) domain restricted by cycle walking is a bijection of [0, n).  Used for the epoch shuffle of an agent update's rollout
// rows (skrl memory.sample_all: shuffle all indices, split into mini-batches; third-party, parity unpinned): minibatch m of an
// epoch takes positions [m * per, m * per + batch) of the permutation, i.e. `batch` DISTINCT rows, and two minibatches of one epoch
// never share a row.  Bit-exact restatement: oracle/rng.py::feistel_permutation.
__device__ __forceinline__ uint32_t feistel_round(uint32_t r, uint32_t k0, uint32_t k1) {
  const uint64_t p = (uint64_t)(r ^ k0) * 0xD2511F53ull;
  return (uint32_t)(p >> 32) ^ (uint32_t)p ^ k1;
}
__device__ __forceinline__ int64_t feistel_permute(int64_t x, int64_t n, int hb, uint64_t seed, uint64_t epoch) {
  const uint32_t mask = (1u << hb) - 1u;
  uint64_t v = (uint64_t)x;
  do {
    uint32_t L = (uint32_t)(v >> hb) & mask, R = (uint32_t)v & mask;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const uint32_t k0 = (uint32_t)seed + 0x9E3779B9u * (uint32_t)(rd + 1), k1 = (uint32_t)(seed >> 32) ^ (uint32_t)epoch ^ (0xBB67AE85u * (uint32_t)(rd + 1)) ^ (uint32_t)(epoch >> 32);
      const uint32_t t = L ^ (feistel_round(R, k0, k1) & mask);
      L = R;
      R = t;
    }
    v = ((uint64_t)L << hb) | R;
  } while ((int64_t)v >= n);
  return (int64_t)v;
}
__global__ __launch_bounds__(kBlock) void rows_take_permuted_kernel(const float* __restrict__ rows, int64_t n, int64_t stride, int dim, int hb,
                                                                    uint64_t seed, uint64_t epoch, int64_t first, int64_t count,
                                                                    float* __restrict__ out, int64_t out_stride, int64_t* __restrict__ idx_out) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= count * dim) return;
  const int64_t i = e / dim;
  const int c = (int)(e - i * dim);
  const int64_t src = feistel_permute(first + i, n, hb, seed, epoch);
  out[i * out_stride + c] = rows[src * stride + c];
  if (idx_out && c == 0) idx_out[i] = src;
}

}  // namespace amp

using namespace amp;

extern "C" {

int amp_ring_create(int64_t capacity, int32_t row_dim, AmpRing** out) {
  AMP_REQUIRE(out, "amp_ring_create: null argument");
  AMP_REQUIRE(capacity >= 1 && capacity < ((int64_t)1 << 32), "amp_ring_create: capacity must be in [1, 2^32)");
  AMP_REQUIRE(row_dim >= 1, "amp_ring_create: row_dim must be positive");
  AmpRing* r = new (std::nothrow) AmpRing();
  AMP_REQUIRE(r, "amp_ring_create: out of host memory");
  r->capacity = capacity;
  r->dim = row_dim;
  r->head = r->size = 0;
  r->rows = nullptr;
  const hipError_t e = hipMalloc(&r->rows, sizeof(float) * (size_t)capacity * row_dim);
  if (e != hipSuccess) {
    delete r;
    return fail(AMP_ERR_HIP, "amp_ring_create: %s", hipGetErrorString(e));
  }
  *out = r;
  return AMP_OK;
}

int amp_ring_destroy(AmpRing* r) {
  if (!r) return AMP_OK;
  (void)hipFree(r->rows);
  delete r;
  return AMP_OK;
}

int64_t amp_ring_size(const AmpRing* r) { return r ? r->size : -1; }
int64_t amp_ring_head(const AmpRing* r) { return r ? r->head : -1; }

int amp_ring_append(AmpRing* r, const float* rows_dev, int64_t n, int64_t row_stride, amp_stream_t stream) {
  AMP_REQUIRE(r, "amp_ring_append: null handle");
  AMP_REQUIRE(n >= 0, "amp_ring_append: negative row count");
  if (n == 0) return AMP_OK;
  AMP_REQUIRE(rows_dev, "amp_ring_append: null rows");
  AMP_REQUIRE(row_stride >= r->dim, "amp_ring_append: row_stride %lld < row_dim %d", (long long)row_stride, r->dim);
  // skrl writes a batch larger than the memory in passes; the net effect is that only the last `capacity` rows
  // survive: skip what would be overwritten within this same call
  int64_t skip = n > r->capacity ? n - r->capacity : 0;
  const int64_t m = n - skip;
  const int64_t head = (r->head + skip) % r->capacity;
  hipStream_t st = (hipStream_t)stream;
  {
    amp::TraceScope trace__("ring_append_kernel", st);
    ring_append_kernel<<<(unsigned)((m * r->dim + kBlock - 1) / kBlock), kBlock, 0, st>>>(rows_dev + skip * row_stride, m, row_stride, r->dim,
                                                                                     r->rows, head, r->capacity);
  }
  const int rc = launch_status("ring_append_kernel");
  if (rc != AMP_OK) return rc;
  r->head = (r->head + n) % r->capacity;
  r->size = r->size + n < r->capacity ? r->size + n : r->capacity;
  return AMP_OK;
}

int amp_rows_take_permuted(const float* rows_dev, int64_t n_rows, int64_t row_stride, int32_t row_dim, uint64_t seed, uint64_t epoch, int64_t first,
                           int64_t count, float* out_dev, int64_t out_stride, int64_t* indices_dev, amp_stream_t stream) {
  AMP_REQUIRE(count >= 0 && first >= 0, "amp_rows_take_permuted: negative range");
  if (count == 0) return AMP_OK;
  AMP_REQUIRE(rows_dev && out_dev, "amp_rows_take_permuted: null buffer");
  AMP_REQUIRE(n_rows >= 1 && n_rows < ((int64_t)1 << 40), "amp_rows_take_permuted: n_rows must be in [1, 2^40)");
  AMP_REQUIRE(first + count <= n_rows, "amp_rows_take_permuted: positions [%lld, %lld) exceed the %lld rows", (long long)first,
              (long long)(first + count), (long long)n_rows);
  AMP_REQUIRE(row_dim >= 1 && row_stride >= row_dim && out_stride >= row_dim, "amp_rows_take_permuted: bad row_dim / strides");
  int bits = 1;
  while (((int64_t)1 << bits) < n_rows) ++bits;
  const int hb = (bits + 1) / 2;   // half-width: 2^(2 hb) >= n_rows
  hipStream_t st = (hipStream_t)stream;
  {
    amp::TraceScope trace__("rows_take_permuted_kernel", st);
    rows_take_permuted_kernel<<<(unsigned)((count * row_dim + kBlock - 1) / kBlock), kBlock, 0, st>>>(rows_dev, n_rows, row_stride, row_dim, hb, seed, epoch,
                                                                                                  first, count, out_dev, out_stride, indices_dev);
  }
  return launch_status("rows_take_permuted_kernel");
}

int amp_ring_sample(const AmpRing* r, uint64_t seed, uint64_t draw, int64_t first_row, int64_t n, float* out_dev, int64_t out_stride,
                    int64_t* indices_dev, amp_stream_t stream) {
  AMP_REQUIRE(r, "amp_ring_sample: null handle");
  AMP_REQUIRE(n >= 0, "amp_ring_sample: negative row count");
  AMP_REQUIRE(first_row >= 0, "amp_ring_sample: negative first_row");
  if (n == 0) return AMP_OK;
  AMP_REQUIRE(r->size > 0, "amp_ring_sample: the buffer is empty");
  AMP_REQUIRE(out_dev, "amp_ring_sample: null output");
  AMP_REQUIRE(out_stride >= r->dim, "amp_ring_sample: out_stride %lld < row_dim %d", (long long)out_stride, r->dim);
  hipStream_t st = (hipStream_t)stream;
  {
    amp::TraceScope trace__("ring_sample_kernel", st);
    ring_sample_kernel<<<(unsigned)((n * r->dim + kBlock - 1) / kBlock), kBlock, 0, st>>>(r->rows, r->size, r->dim, seed, draw, first_row, n, out_dev,
                                                                                     out_stride, indices_dev);
  }
  return launch_status("ring_sample_kernel");
}

}  // extern "C"
