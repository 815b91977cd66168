// Device body of the reset-id compaction (compact.hip), shared with the fused step tail of disc.hip.
#pragma once
#include "amp_common.hpp"

namespace amp {

constexpr int kTile = 64;

// counts[] holds one entry per `64 / sub` envs (sub = 1, 2, 4 or 8 count entries per 64-env wave tile).  `block` = index of
// the 4-tile group this 256-thread workgroup owns.
__device__ __forceinline__ void compact_scatter_body(const int64_t block, const uint8_t* __restrict__ mask,
                                                     const int32_t* __restrict__ counts, int64_t N, int64_t n_tiles, int sub,
                                                     int64_t n_counts, int64_t* __restrict__ ids,
                                                     int64_t* __restrict__ count_out) {
  __shared__ long long s_part[kBlock / kWave];
  __shared__ long long s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t first_tile = block * (kBlock / kTile);
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
