// Device body of the reset-id compaction (compact.hip), shared with the fused step tail of disc.hip.
#pragma once
#include "amp_common.hpp"

namespace amp {

constexpr int kTile = 64;

// counts[] holds one entry per `64 / sub` envs (sub = 1, 2, 4 or 8 count entries per 64-env wave tile).  `block` = index of
// the 4-tile group this 256-thread workgroup owns.
// Returns this lane's slot in the ascending id list (-1: its env is not reset); `env_out` = the env the lane looks at.
__device__ __forceinline__ int64_t compact_rank_body(const int64_t block, const uint8_t* __restrict__ mask,
                                                     const int32_t* __restrict__ counts, int64_t N, int64_t n_tiles, int sub,
                                                     int64_t n_counts, int64_t* __restrict__ ids,
                                                     int64_t* __restrict__ count_out, int64_t& env_out) {
  __shared__ long long s_part[kBlock / kWave];
  __shared__ long long s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t first_tile = block * (kBlock / kTile);
  // exclusive prefix of the tile counts before this workgroup
  long long acc = 0;
  {
    // The entries before this workgroup (a multiple of four): eight independent 16-B loads per trip cover 8 192 entries,
    // i.e. every case up to 65 536 envs at 8-env tiles, in ONE round trip.  One 4-B load per trip was a chain of up to 32
    // dependent misses in the last workgroups (the counts were just written by other XCDs: 14 us of tail at 8-env tiles).
    const int64_t n_before = first_tile * sub;
    if ((reinterpret_cast<uintptr_t>(counts) & 15) == 0) {
      typedef int int4v __attribute__((ext_vector_type(4)));
      const int4v* c4 = reinterpret_cast<const int4v*>(counts);
      const int64_t n4 = n_before >> 2;  // first_tile is a multiple of four
      for (int64_t t = tid; t < n4; t += 8 * kBlock) {
        int4v v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t + k * kBlock < n4 ? c4[t + k * kBlock] : int4v{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += (long long)v[k][0] + v[k][1] + v[k][2] + v[k][3];
      }
    } else {
      for (int64_t t = tid; t < n_before; t += kBlock) acc += counts[t];
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) s_part[wave] = acc;
  __syncthreads();
  if (tid == 0) s_base = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
  __syncthreads();
  long long base = s_base;
  const int64_t my_tile = first_tile + wave;
  {
    // counts of this workgroup's earlier tiles (at most 3 tiles x 8 entries): one entry per lane, summed across the wave --
    // a serial loop here was a chain of up to 24 dependent L2 round trips (17 us of tail at 8-env tiles)
    const int64_t c = first_tile * sub + lane;
    long long part = (c < my_tile * sub && c < n_counts) ? counts[c] : 0;
    for (int off = 16; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    base += __shfl(part, 0, 64);
  }
  const int64_t i = my_tile * kTile + lane;
  const int bit = (i < N) ? (mask[i] != 0) : 0;
  const unsigned long long b = __ballot(bit);
  int64_t slot = -1;
  if (bit) {
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
    slot = base + rank;
    ids[slot] = i;
  }
  // the workgroup holding the last tile publishes the total
  if (my_tile == n_tiles - 1 && lane == 0) *count_out = base + __popcll(b);
  env_out = i;
  return slot;
}

__device__ __forceinline__ void compact_scatter_body(const int64_t block, const uint8_t* __restrict__ mask,
                                                     const int32_t* __restrict__ counts, int64_t N, int64_t n_tiles, int sub,
                                                     int64_t n_counts, int64_t* __restrict__ ids,
                                                     int64_t* __restrict__ count_out) {
  int64_t env;
  (void)compact_rank_body(block, mask, counts, N, n_tiles, sub, n_counts, ids, count_out, env);
}

}  // namespace amp
