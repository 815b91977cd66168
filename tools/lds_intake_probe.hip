// How fast can ONE CU take operand bytes in?  The small-shard layer-2 tile (128 x 128, 1 MiB of operand fills per tile, one
// tile per CU) sits at ~28 us against 12-16 us of MFMAs; the round-2 fill probe (tools/gemm_f16_bench.hip FILL=1: 17.1 TB/s
// chip-wide with full-line pieces = ~32 B / clk / CU) waited for vmcnt(0) + a barrier per ring slot, so it may have measured
// latency, not the intake ceiling.  This probe streams with a COUNTED wait and no barrier:
//   PATH 0: LDS-DMA (global_load_lds, 16 B per lane = 1 KiB per wave instruction), DEPTH pieces in flight per wave;
//   PATH 1: global_load_dwordx4 into registers + ds_write_b128, DEPTH loads in flight per wave;
//   PATH 2: both at once (even waves LDS-DMA, odd waves the register path).
// Sources: a 2 MB buffer every workgroup reads (L2-resident, like the layer-2 weights), a 32 MB buffer read in 1-MiB spans, eight
// workgroups per span (Infinity-Cache-resident, like an 8 192-row hidden layer), a 1 GB buffer (HBM).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_intake_probe.hip -o tools/bin/lds_intake_probe && tools/bin/lds_intake_probe
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

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned uv4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Each wave streams `pieces` 1-KiB pieces: piece i of wave w of workgroup b comes from
//   src + ((b * span_stride + (w * pieces + i) * 1024) % wrap) + lane * 16
// (wrap = bytes of the source that are actually touched; span_stride = 0: every workgroup reads the same bytes).
template <int PATH, int DEPTH, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void intake_kernel(const unsigned char* src, int64_t span_stride, int64_t wrap, int pieces,
                                                            float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* ring = lds + wave * (DEPTH * 1024);
  const int64_t base = (int64_t)blockIdx.x * span_stride + (int64_t)wave * pieces * 1024;
  auto addr = [&](int i) { return src + (base + (int64_t)i * 1024) % wrap + lane * 16; };
  const bool dma = PATH == 0 || (PATH == 2 && (wave & 1) == 0);
  if (dma) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) __builtin_amdgcn_global_load_lds((gptr_t)addr(i), (lptr_t)(ring + i * 1024), 16, 0, 0);
    for (int i = DEPTH; i < pieces; i += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        wait_vm<DEPTH - 1>();  // the piece that used this slot has landed
        __builtin_amdgcn_global_load_lds((gptr_t)addr(i + j), (lptr_t)(ring + j * 1024), 16, 0, 0);
      }
    }
    wait_vm<0>();
  } else {
    uv4 v[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const uv4*>(addr(i)));
    for (int i = DEPTH; i < pieces; i += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        wait_vm<DEPTH - 1>();
        *reinterpret_cast<uv4*>(ring + j * 1024 + lane * 16) = v[j];
        v[j] = __builtin_nontemporal_load(reinterpret_cast<const uv4*>(addr(i + j)));
      }
    }
    wait_vm<0>();
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) *reinterpret_cast<uv4*>(ring + j * 1024 + lane * 16) = v[j];
  }
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = reinterpret_cast<float*>(lds)[lane];
}

template <int PATH, int DEPTH, int THREADS>
static void run(const char* what, const unsigned char* src, int64_t span_stride, int64_t wrap, int pieces, float* sink) {
  const int lds = (THREADS / 64) * DEPTH * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(intake_kernel<PATH, DEPTH, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) intake_kernel<PATH, DEPTH, THREADS><<<256, THREADS, lds>>>(src, span_stride, wrap, pieces, sink);
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) intake_kernel<PATH, DEPTH, THREADS><<<256, THREADS, lds>>>(src, span_stride, wrap, pieces, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / reps;
  const double bytes = 256.0 * (THREADS / 64) * pieces * 1024.0;
  printf("%-28s path %d depth %2d waves %d: %7.1f us  %6.2f TB/s  %6.1f GB/s per CU  (%.0f KiB per CU)\n", what, PATH, DEPTH, THREADS / 64, us,
         bytes / (us * 1e-6) / 1e12, bytes / 256 / (us * 1e-6) / 1e9, bytes / 256 / 1024);
  fflush(stdout);
}

int main() {
  const int64_t big = (int64_t)1 << 30;
  unsigned char* buf;
  float* sink;
  CK(hipMalloc(&buf, big));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(buf, 1, big));
  // 1 MiB per CU (= one 128 x 128 layer-2 tile's fills) unless stated
  auto sweep = [&](const char* what, int64_t span_stride, int64_t wrap, int kib_per_cu) {
    const int p4 = kib_per_cu / 4, p8 = kib_per_cu / 8;
    run<0, 4, 256>(what, buf, span_stride, wrap, p4, sink);
    run<0, 8, 256>(what, buf, span_stride, wrap, p4, sink);
    run<0, 16, 256>(what, buf, span_stride, wrap, p4, sink);
    run<0, 8, 512>(what, buf, span_stride, wrap, p8, sink);
    run<0, 16, 512>(what, buf, span_stride, wrap, p8, sink);
    run<1, 4, 256>(what, buf, span_stride, wrap, p4, sink);
    run<1, 8, 256>(what, buf, span_stride, wrap, p4, sink);
    run<1, 16, 256>(what, buf, span_stride, wrap, p4, sink);
    run<1, 8, 512>(what, buf, span_stride, wrap, p8, sink);
    run<1, 16, 512>(what, buf, span_stride, wrap, p8, sink);
    run<2, 8, 512>(what, buf, span_stride, wrap, p8, sink);
    run<2, 16, 512>(what, buf, span_stride, wrap, p8, sink);
  };
  sweep("2 MB shared (L2)", 0, (int64_t)2 << 20, 1024);
  sweep("32 MB, 8 CUs per MiB (MALL)", (int64_t)1 << 20, (int64_t)32 << 20, 1024);
  sweep("2 MB shared (L2), 8 MiB", 0, (int64_t)2 << 20, 8192);  // steady state: launch and ramp amortised
  sweep("32 MB (MALL), 8 MiB", (int64_t)1 << 20, (int64_t)32 << 20, 8192);
  sweep("1 GB disjoint (HBM)", (int64_t)4 << 20, big, 4096);
  return 0;
}
