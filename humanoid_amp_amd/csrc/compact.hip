// Reset-index compaction: mask[N] -> ascending int64 ids + count, bit-exact with
// reset_buf.nonzero(as_tuple=False).squeeze(-1) (DirectRLEnv.step; consumer g1_amp_env.py:332-358).
//
// Deterministic two-level scan, no inter-workgroup hand-off inside a launch:
//   pass 1 (skipped when amp_env_step already produced them): per-64-env tile counts via wave ballot + popcount
//   pass 2: a workgroup owns 4 tiles (one per wave).  Its base offset is the sum of all earlier tile counts
//           (block-wide reduction of <= N/64 ints, L2 resident); inside a wave the rank of a set lane is
//           mbcnt(ballot) -- gfx950 v_mbcnt_lo/hi -- so ids come out ascending by construction.
#include "amp_common.hpp"

namespace amp {

constexpr int kTile = 64;

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

// counts[] holds one entry per `64 / sub` envs (sub = 1, 2 or 4 count entries per 64-env wave tile)
__global__ __launch_bounds__(kBlock) void compact_scatter_kernel(const uint8_t* __restrict__ mask,
                                                                 const int32_t* __restrict__ counts, int64_t N,
                                                                 int64_t n_tiles, int sub, int64_t n_counts,
                                                                 int64_t* __restrict__ ids,
                                                                 int64_t* __restrict__ count_out) {
  __shared__ long long s_part[kBlock / kWave];
  __shared__ long long s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t first_tile = (int64_t)blockIdx.x * (kBlock / kTile);
  // exclusive prefix of the tile counts before this workgroup
  long long acc = 0;
  for (int64_t t = tid; t < first_tile * sub; t += kBlock) acc += counts[t];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) s_part[wave] = acc;
  __syncthreads();
  if (tid == 0) s_base = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
  __syncthreads();
  long long base = s_base;
  const int64_t my_tile = first_tile + wave;
  for (int64_t c = first_tile * sub; c < my_tile * sub && c < n_counts; ++c) base += counts[c];
  const int64_t i = my_tile * kTile + lane;
  const int bit = (i < N) ? (mask[i] != 0) : 0;
  const unsigned long long b = __ballot(bit);
  if (bit) {
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
    ids[base + rank] = i;
  }
  // the workgroup holding the last tile publishes the total
  if (my_tile == n_tiles - 1 && lane == 0) *count_out = base + __popcll(b);
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
  AMP_REQUIRE(tile_envs == 16 || tile_envs == 32 || tile_envs == 64, "amp_reset_compact: tile_envs must be 16, 32 or 64");
  const int sub = kTile / tile_envs;
  const int64_t n_counts = (N + tile_envs - 1) / tile_envs;
  const int64_t n_tiles = (N + kTile - 1) / kTile;
  const unsigned grid = (unsigned)((n_tiles + 3) / 4);
  { amp::TraceScope trace__("compact_scatter_kernel", (hipStream_t)stream);
    compact_scatter_kernel<<<grid, kBlock, 0, (hipStream_t)stream>>>(mask, counts, N, n_tiles, sub, n_counts, ids, count);
  }
  return launch_status("compact_scatter_kernel");
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
