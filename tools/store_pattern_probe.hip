// What does the store PATTERN of layer 1's epilogue cost?  The hidden layer is [M rows][4 096 B]; a 512-thread workgroup owns a
// 256-row x 1 024-B tile (256 columns x 4 B) as 8 waves of 128 rows x 256 B, and writes it with
//   PAT 0: buffer_store_dword, one instruction = two 128-B row segments (lanes 0-31 row r, lanes 32-63 row r + 4): the product
//          epilogue's pattern, 128 instructions per wave;
//   PAT 1: global_store_dwordx2, one instruction = four 128-B segments (16 lanes x 8 B each), 64 instructions per wave;
//   PAT 2: global_store_dwordx4, one instruction = eight 128-B segments (8 lanes x 16 B each), 32 instructions per wave;
//   PAT 3: global_store_dwordx4, one instruction = four 256-B row segments (16 lanes x 16 B), 32 instructions per wave.
// No arithmetic, no loads: the bytes per second of each pattern at M = 32 768 (134 MB) and M = 8 192 (33.5 MB).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_pattern_probe.hip -o tools/bin/store_pattern_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("%s: %s\n", #x, hipGetErrorString(e));                            \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef unsigned uv2 __attribute__((ext_vector_type(2)));
typedef unsigned uv4 __attribute__((ext_vector_type(4)));

template <int PAT, int NT>
__global__ __launch_bounds__(512, 1) void store_kernel(unsigned char* out, int m_tiles) {
  constexpr int kPitch = 4096;
  const int per_xcd = (m_tiles * 4 + 7) / 8;
  const int v = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (v >= m_tiles * 4) return;
  const int mt = v >> 2, nt = v & 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  unsigned char* base = out + ((int64_t)mt * 256 + wm * 128) * kPitch + nt * 1024 + wn * 256;  // the wave's 128 rows x 256 B
  const unsigned x = tid * 2654435761u;
  if (PAT == 0) {
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          unsigned char* p = base + (int64_t)(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * kPitch + b * 128 + li * 4;
          if (NT) __builtin_nontemporal_store(x + r, reinterpret_cast<unsigned*>(p));
          else *reinterpret_cast<unsigned*>(p) = x + r;
        }
  } else if (PAT == 1) {
    const int c = lane & 15, rr = lane >> 4;  // 16 lanes per 128-B segment, 4 rows per instruction
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        unsigned char* p = base + (int64_t)(i * 4 + rr) * kPitch + b * 128 + c * 8;
        const uv2 w = {x + i, x};
        if (NT) __builtin_nontemporal_store(w, reinterpret_cast<uv2*>(p));
        else *reinterpret_cast<uv2*>(p) = w;
      }
  } else if (PAT == 2) {
    const int c = lane & 7, rr = lane >> 3;  // 8 lanes per 128-B segment, 8 rows per instruction
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        unsigned char* p = base + (int64_t)(i * 8 + rr) * kPitch + b * 128 + c * 16;
        const uv4 w = {x + i, x, x, x};
        if (NT) __builtin_nontemporal_store(w, reinterpret_cast<uv4*>(p));
        else *reinterpret_cast<uv4*>(p) = w;
      }
  } else {
    const int c = lane & 15, rr = lane >> 4;  // 16 lanes per 256-B row segment, 4 rows per instruction
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      unsigned char* p = base + (int64_t)(i * 4 + rr) * kPitch + c * 16;
      const uv4 w = {x + i, x, x, x};
      if (NT) __builtin_nontemporal_store(w, reinterpret_cast<uv4*>(p));
      else *reinterpret_cast<uv4*>(p) = w;
    }
  }
}

template <int PAT, int NT>
static void run(unsigned char* out, int64_t M) {
  const int m_tiles = (int)(M / 256);
  const unsigned grid = (unsigned)((m_tiles * 4 + 7) / 8 * 8);
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) store_kernel<PAT, NT><<<grid, 512>>>(out, m_tiles);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) store_kernel<PAT, NT><<<grid, 512>>>(out, m_tiles);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / reps, bytes = (double)M * 4096;
  printf("M = %6lld  pattern %d%s: %7.1f us  %5.2f TB/s\n", (long long)M, PAT, NT ? " nt" : "   ", us, bytes / (us * 1e-6) / 1e12);
  fflush(stdout);
}

int main() {
  unsigned char* out;
  CK(hipMalloc(&out, (size_t)65536 * 4096));
  for (int64_t M : {32768, 8192, 65536}) {
    run<0, 0>(out, M);
    run<1, 0>(out, M);
    run<2, 0>(out, M);
    run<3, 0>(out, M);
    run<0, 1>(out, M);
    run<2, 1>(out, M);
    run<3, 1>(out, M);
  }
  return 0;
}
