// Reset-index compaction: mask[N] -> ascending int64 ids + count, bit-exact with
// reset_buf.nonzero(as_tuple=False).squeeze(-1) (DirectRLEnv.step; consumer g1_amp_env.py:332-358).
//
// Deterministic two-level scan, no inter-workgroup hand-off inside a launch:
//   pass 1 (skipped when amp_env_step already produced them): per-64-env tile counts via wave ballot + popcount
//   pass 2: a workgroup owns 4 tiles (one per wave).  Its base offset is the sum of all earlier tile counts
//           (block-wide reduction of <= N/64 ints, L2 resident); inside a wave the rank of a set lane is
//           mbcnt(ballot) -- gfx950 v_mbcnt_lo/hi -- so ids come out ascending by construction.
#include "compact_kernels.hpp"

namespace amp {

__global__ __launch_bounds__(kBlock) void tile_count_kernel(const uint8_t* __restrict__ mask, int64_t N,
                                                            int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int bit = (i < N) ? (mask[i] != 0) : 0;
  const unsigned long long b = __ballot(bit);
  if ((threadIdx.x & 63) == 0) {
    const int64_t tile = i / kTile;
    if (tile * kTile < N) counts[tile] = __popcll(b);
  }
}

__global__ __launch_bounds__(kBlock) void compact_scatter_kernel(const uint8_t* __restrict__ mask,
                                                                 const int32_t* __restrict__ counts, int64_t N,
                                                                 int64_t n_tiles, int sub, int64_t n_counts,
                                                                 int64_t* __restrict__ ids,
                                                                 int64_t* __restrict__ count_out) {
  compact_scatter_body(blockIdx.x, mask, counts, N, n_tiles, sub, n_counts, ids, count_out);
}

// Device-count-bounded row scatter (amp_scatter_rows): item (i, r), i < min(max_n, *count), r < repeat, is one workgroup
// pass over `width` floats.  Up to kMaxScatterOps independent ops share a launch (blockIdx.y = op).
constexpr int kMaxScatterOps = 8;
struct ScatterOps {
  AmpScatterRows op[kMaxScatterOps];
};
__global__ __launch_bounds__(kBlock) void scatter_rows_kernel(ScatterOps ops, const int64_t* __restrict__ ids,
                                                              const int64_t* __restrict__ count, int64_t max_n) {
  const AmpScatterRows& o = ops.op[blockIdx.y];
  const int64_t n = *count < max_n ? *count : max_n;
  const int per = o.width * o.repeat;  // floats per destination row
  // a thread owns one float of one destination row; rows are walked with a grid stride (n is only known on the device)
  const int64_t total = n * per;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t i = e / per;
    const int c = (int)(e - i * per), w = c % o.width;
    float v = o.src ? o.src[i * o.src_stride + w] : o.fill;
    if (o.add) v += o.add[c];
    o.dst[ids[i] * o.dst_stride + c] = v;
  }
}

}  // namespace amp

using namespace amp;

extern "C" {

int64_t amp_reset_compact_workspace_bytes(int64_t N) {
  if (N < 0) return -1;
  return (int64_t)sizeof(int32_t) * ((N + kTile - 1) / kTile + 1);
}

int amp_reset_compact_tiles(const uint8_t* mask, const int32_t* counts, int32_t tile_envs, int64_t N, int64_t* ids,
                            int64_t* count, amp_stream_t stream) {
  AMP_REQUIRE(N >= 0, "amp_reset_compact: negative num_envs");
  AMP_REQUIRE(count, "amp_reset_compact: count pointer is null");
  if (N == 0) {
    AMP_HIP(hipMemsetAsync(count, 0, sizeof(int64_t), (hipStream_t)stream));
    return AMP_OK;
  }
  AMP_REQUIRE(mask && counts && ids, "amp_reset_compact: null buffer");
  AMP_REQUIRE(tile_envs == 8 || tile_envs == 16 || tile_envs == 32 || tile_envs == 64, "amp_reset_compact: tile_envs must be 8, 16, 32 or 64");
  const int sub = kTile / tile_envs;
  const int64_t n_counts = (N + tile_envs - 1) / tile_envs;
  const int64_t n_tiles = (N + kTile - 1) / kTile;
  const unsigned grid = (unsigned)((n_tiles + 3) / 4);
  { amp::TraceScope trace__("compact_scatter_kernel", (hipStream_t)stream);
    compact_scatter_kernel<<<grid, kBlock, 0, (hipStream_t)stream>>>(mask, counts, N, n_tiles, sub, n_counts, ids, count);
  }
  return launch_status("compact_scatter_kernel");
}

int amp_scatter_rows(const AmpScatterRows* ops, int32_t n_ops, const int64_t* ids, const int64_t* count, int64_t max_n,
                     amp_stream_t stream) {
  AMP_REQUIRE(n_ops >= 0 && n_ops <= kMaxScatterOps, "amp_scatter_rows: at most %d ops per call", kMaxScatterOps);
  AMP_REQUIRE(max_n >= 0, "amp_scatter_rows: negative max_n");
  if (n_ops == 0 || max_n == 0) return AMP_OK;
  AMP_REQUIRE(ops && ids && count, "amp_scatter_rows: null argument");
  ScatterOps k{};
  int per_max = 1;
  for (int i = 0; i < n_ops; ++i) {
    const AmpScatterRows& o = ops[i];
    AMP_REQUIRE(o.dst && o.width >= 1 && o.repeat >= 1, "amp_scatter_rows: op %d needs dst, width >= 1 and repeat >= 1", i);
    AMP_REQUIRE(o.dst_stride >= (int64_t)o.width * o.repeat, "amp_scatter_rows: op %d: dst_stride smaller than width * repeat", i);
    AMP_REQUIRE(!o.src || o.src_stride >= o.width, "amp_scatter_rows: op %d: src_stride smaller than width", i);
    k.op[i] = o;
    per_max = o.width * o.repeat > per_max ? o.width * o.repeat : per_max;
  }
  // enough workgroups for a few hundred rows per pass; the grid-stride loop covers any count
  int64_t want = (max_n * per_max + kBlock - 1) / kBlock;
  const unsigned gx = (unsigned)(want < 1 ? 1 : (want > 256 ? 256 : want));
  { amp::TraceScope trace__("scatter_rows_kernel", (hipStream_t)stream);
    scatter_rows_kernel<<<dim3(gx, (unsigned)n_ops), kBlock, 0, (hipStream_t)stream>>>(k, ids, count, max_n);
  }
  return launch_status("scatter_rows_kernel");
}

int amp_reset_compact(const uint8_t* mask, int64_t N, int64_t* ids, int64_t* count, void* workspace, amp_stream_t stream) {
  AMP_REQUIRE(N >= 0, "amp_reset_compact: negative num_envs");
  if (N > 0) {
    AMP_REQUIRE(mask && workspace, "amp_reset_compact: null buffer");
    const unsigned grid = (unsigned)((N + kBlock - 1) / kBlock);
    { amp::TraceScope trace__("tile_count_kernel", (hipStream_t)stream);
      tile_count_kernel<<<grid, kBlock, 0, (hipStream_t)stream>>>(mask, N, (int32_t*)workspace);
    }
    int rc = launch_status("tile_count_kernel");
    if (rc != AMP_OK) return rc;
  }
  return amp_reset_compact_tiles(mask, (const int32_t*)workspace, kTile, N, ids, count, stream);
}

}  // extern "C"
