// Shared host/device helpers of libamp_engine.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/amp_engine.h"

namespace amp {

// ---- error plumbing: no exception crosses the C ABI ------------------------------------------------
char* last_error_buf();  // thread-local, 512 bytes
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define AMP_HIP(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) return ::amp::fail(AMP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define AMP_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return ::amp::fail(AMP_ERR_INVALID, __VA_ARGS__); \
  } while (0)

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AMP_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return AMP_OK;
}

// ---- kernel tracer (core.hip) ----------------------------------------------------------------------
bool trace_enabled();
int trace_open(const char* kernel, hipStream_t st);  // -> record index or -1
void trace_close(int rec, hipStream_t st);
struct TraceScope {
  int rec;
  hipStream_t st;
  TraceScope(const char* kernel, hipStream_t s) : rec(trace_enabled() ? trace_open(kernel, s) : -1), st(s) {}
  ~TraceScope() {
    if (rec >= 0) trace_close(rec, st);
  }
};

constexpr int kWave = 64;     // gfx950 wavefront
constexpr int kBlock = 256;   // 4 waves, one per SIMD
constexpr int kMaxKey = 8;

// ---- device-side math restating the reference's fp32 op order (compiled with -ffp-contract=off) ----
struct Quat {
  float w, x, y, z;
};

__device__ __forceinline__ float lerp_ref(float a, float b, float t) {
  // (1 - t) * a + t * b : sub, mul, mul, add -- motions/motion_loader.py:215
  return (1.0f - t) * a + t * b;
}

// sinf / acosf for the arguments SLERP produces, in ~25 VALU instructions each instead of the ~100 of the general library
// routines (their range reduction for huge arguments and their special cases are most of what a SLERP used to execute:
// MotionLoader.sample was bound by them, not by memory).  fdlibm's float kernels: two-step Cody-Waite reduction by pi/2 with
// fused multiply-adds (|x| <= 64, beyond that the library routine) + the k_sinf / k_cosf polynomials; e_acosf's two branches
// on [0, 1].  Measured against float64 on 2.4 M arguments (numpy emulation of the same float32 operations): sin <= 1.45 ulp
// (glibc sinf: 1.40), acos <= 0.80 ulp (glibc acosf: 2.1).
__device__ __forceinline__ float sin_bounded(float x) {
  if (__builtin_expect(fabsf(x) > 64.0f, 0)) return sinf(x);
  const float n = rintf(x * 0.636619746685028076171875f);  // x * 2 / pi, round half to even
  float r = __builtin_fmaf(-n, 1.57079637050628662109375f, x);
  r = __builtin_fmaf(-n, -4.371139000186241e-08f, r);
  const float z = r * r;
  const float sp = r + (z * r) * (-1.6666667163e-01f + z * (8.3333337680e-03f + z * (-1.9841270114e-04f + z * (2.7557314297e-06f + z * (-2.5050759689e-08f + z * 1.5896910177e-10f)))));
  const float rr = z * (4.1666667908e-02f + z * (-1.3888889225e-03f + z * (2.4801587642e-05f + z * (-2.7557314297e-07f + z * (2.0875723372e-09f + z * -1.1359647598e-11f)))));
  const float cp = 1.0f - (0.5f * z - z * rr);
  const int k = (int)n;
  const float v = (k & 1) ? cp : sp;
  return (k & 2) ? -v : v;
}

__device__ __forceinline__ float acos_unit(float x) {  // x in [0, 1]; > 1 gives NaN like acosf (masked by SLERP's last branch)
  const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
  if (x < 0.5f) {
    const float z = x * x;
    const float r = (z * (pS0 + z * (pS1 + z * pS2))) / (1.0f + z * qS1);
    return 1.5707962513e+00f - (x - (7.5497894159e-08f - x * r));
  }
  const float z = (1.0f - x) * 0.5f;
  const float s = sqrtf(z);
  const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
  const float c = s > 0.0f ? (z - df * df) / (s + df) : 0.0f;
  const float r = (z * (pS0 + z * (pS1 + z * pS2))) / (1.0f + z * qS1);
  return 2.0f * (df + (r * s + c));
}

__device__ __forceinline__ Quat slerp_ref(Quat q0, Quat q1, float t) {
  // motions/motion_loader.py:247-279 (wxyz; no renormalisation; branch order matters)
  float c = ((q0.w * q1.w + q0.x * q1.x) + q0.y * q1.y) + q0.z * q1.z;
  if (c < 0.0f) {
    q1.w = -q1.w;
    q1.x = -q1.x;
    q1.y = -q1.y;
    q1.z = -q1.z;
  }
  c = fabsf(c);
  const float half = acos_unit(c);
  const float s = sqrtf(1.0f - c * c);
  const float ra = sin_bounded((1.0f - t) * half) / s;
  const float rb = sin_bounded(t * half) / s;
  Quat o;
  o.w = ra * q0.w + rb * q1.w;
  o.x = ra * q0.x + rb * q1.x;
  o.y = ra * q0.y + rb * q1.y;
  o.z = ra * q0.z + rb * q1.z;
  if (fabsf(s) < 0.001f) {
    o.w = 0.5f * q0.w + 0.5f * q1.w;
    o.x = 0.5f * q0.x + 0.5f * q1.x;
    o.y = 0.5f * q0.y + 0.5f * q1.y;
    o.z = 0.5f * q0.z + 0.5f * q1.z;
  }
  if (fabsf(c) >= 1.0f) o = q0;
  return o;
}

struct Vec3 {
  float x, y, z;
};

__device__ __forceinline__ Vec3 cross_ref(Vec3 a, Vec3 b) {
  return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

__device__ __forceinline__ Vec3 quat_apply_ref(Quat q, Vec3 v) {
  // Isaac Lab quat_apply: t = 2 * (q_xyz x v);  v + w * t + q_xyz x t   (call sites g1_amp_env.py:495-496)
  const Vec3 u{q.x, q.y, q.z};
  Vec3 t = cross_ref(u, v);
  t.x *= 2.0f;
  t.y *= 2.0f;
  t.z *= 2.0f;
  const Vec3 c = cross_ref(u, t);
  return Vec3{(v.x + q.w * t.x) + c.x, (v.y + q.w * t.y) + c.y, (v.z + q.w * t.z) + c.z};
}

__device__ __forceinline__ Vec3 quat_rotate_inverse_ref(Quat q, Vec3 v) {
  // Isaac Lab quat_rotate_inverse: a - b + c  (call site g1_amp_env.py:253)
  const float s = 2.0f * (q.w * q.w) - 1.0f;
  const Vec3 u{q.x, q.y, q.z};
  const Vec3 cr = cross_ref(u, v);
  const float d = (u.x * v.x + u.y * v.y) + u.z * v.z;
  Vec3 o;
  o.x = (v.x * s - (cr.x * q.w) * 2.0f) + (u.x * d) * 2.0f;
  o.y = (v.y * s - (cr.y * q.w) * 2.0f) + (u.y * d) * 2.0f;
  o.z = (v.z * s - (cr.z * q.w) * 2.0f) + (u.z * d) * 2.0f;
  return o;
}

// ---- Philox4x32-10 (Salmon et al., SC'11; Random123 constants): the engine's counter-based generator; pinned bit for
// bit by oracle/rng.py, itself pinned by the Random123 known-answer vectors
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}


// ---- fp16 plane pairs (disc_gemm_f16.hpp; also written by amp_env_step's fused scaler) -----------------------------
// one (p0, p1) pair per element: p0 = rn16(v), p1 = rn16(v - p0) for a value already multiplied by its plane scale
__device__ __forceinline__ uint32_t plane_pair(float v) {
  const _Float16 a = (_Float16)v;
  const _Float16 b = (_Float16)(v - (float)a);
  return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}

}  // namespace amp
