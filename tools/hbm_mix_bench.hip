// Measurement aid: what this device's memory system sustains on the byte mix of the env-step launch (78 MB read +
// 158 MB written per launch, 16-B accesses) -- the practical ceiling to read roofline_hbm against.  Streams rotate over
// `sets` buffer sets larger than the 256-MB last-level cache.
//   mode 0: read 1 x 16 B, write 2 x 16 B per item (the env-step mix)   1: copy (1 : 1)   2: read only   3: write only
// build: hipcc -O3 --offload-arch=gfx950 tools/hbm_mix_bench.hip -o tools/bin/hbm_mix_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void mix_kernel(const f4* __restrict__ src, f4* __restrict__ dst, long n_items, float* sink) {
  const long stride = (long)gridDim.x * 256;
  f4 accum = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_items; i += stride) {
    f4 v = {1, 2, 3, 4};
    if (MODE != 3) v = src[i];
    if (MODE == 0) { dst[i] = v; dst[n_items + i] = v * 2.0f; }
    else if (MODE == 1 || MODE == 3) dst[i] = v;
    else accum += v;
  }
  if (MODE == 2 && accum[0] == 123.456f) sink[0] = accum[1];
}

template <int MODE>
static void run(long n_items, int grid, int sets, f4** src, f4** dst, float* sink, double bytes) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 4; ++w) mix_kernel<MODE><<<grid, 256>>>(src[w % sets], dst[w % sets], n_items, sink);
  const int reps = 24;
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) mix_kernel<MODE><<<grid, 256>>>(src[r % sets], dst[r % sets], n_items, sink);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d grid %5d: %.1f us per launch, %.2f TB/s\n", MODE, grid, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const double read_mb = argc > 1 ? atof(argv[1]) : 78.0;
  const int sets = argc > 2 ? atoi(argv[2]) : 4;
  const long n_items = (long)(read_mb * 1e6 / 16);
  std::vector<f4*> src(sets), dst(sets);
  for (int s = 0; s < sets; ++s) {
    hipMalloc(&src[s], n_items * 16); hipMalloc(&dst[s], 2 * n_items * 16);
    hipMemset(src[s], 0, n_items * 16); hipMemset(dst[s], 0, 2 * n_items * 16);
  }
  float* sink; hipMalloc(&sink, 16);
  const int grids[] = {1024, 2048, 4096, 8192, 16384};
  for (int g : grids) run<0>(n_items, g, sets, src.data(), dst.data(), sink, 3.0 * n_items * 16);
  for (int g : grids) run<1>(n_items, g, sets, src.data(), dst.data(), sink, 2.0 * n_items * 16);
  for (int g : grids) run<2>(n_items, g, sets, src.data(), dst.data(), sink, 1.0 * n_items * 16);
  for (int g : grids) run<3>(n_items, g, sets, src.data(), dst.data(), sink, 1.0 * n_items * 16);
  return 0;
}
