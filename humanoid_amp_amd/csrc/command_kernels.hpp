// The counter-based velocity-command draw (command.hip), shared with the fused device-side reset (motion.hip).
#pragma once
#include "amp_common.hpp"

namespace amp {

constexpr uint32_t kCommandDomain = 0xA14C0000u;  // xor-ed into the key's high word: tick = +0, reset = +1

__device__ __forceinline__ float u01_24(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-08f; }  // [0, 1), 24 bits

__device__ __forceinline__ void draw_command(uint64_t seed, uint64_t step, uint64_t env, uint32_t mode, float vel_lo,
                                             float vel_span, float t_lo, float t_span, float& cx, float& cy, float& tl) {
  uint32_t r[4];
  philox4x32_10((uint32_t)env, (uint32_t)(env >> 32), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32) ^ (kCommandDomain + mode), r);
  // torch.rand(...) * (hi - lo) + lo: one fp32 multiply, one fp32 add (contraction is off)
  cx = u01_24(r[0]) * vel_span + vel_lo;
  cy = u01_24(r[1]) * vel_span + vel_lo;
  tl = u01_24(r[2]) * t_span + t_lo;
}

// the reset-side resample of one env (g1_amp_env.py:421-439): the ranged draw, or the fixed command (lo, 0) with an
// infinite timer when the range is empty (:436-439)
__device__ __forceinline__ uint64_t command_step_of(const AmpCommandArgs& a) { return a.step + (a.step_dev ? *a.step_dev : 0ull); }

__device__ __forceinline__ void command_reset_env(const AmpCommandArgs& a, int64_t env) {
  if (a.vel_span > 0.0f) {
    float cx, cy, tl;
    draw_command(a.seed, command_step_of(a), (uint64_t)(a.env_offset + env), 1u, a.vel_lo, a.vel_span, a.t_lo, a.t_span, cx, cy, tl);
    a.command[2 * env] = cx;
    a.command[2 * env + 1] = cy;
    a.time_left[env] = tl;
  } else {
    a.command[2 * env] = a.vel_lo;
    a.command[2 * env + 1] = 0.0f;
    a.time_left[env] = __builtin_inff();
  }
}

}  // namespace amp
