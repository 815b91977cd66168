// Velocity-command timers of the G1 env and the lazily read reward-log means.
//
//   command_kernel   G1AmpEnv._pre_physics_step (g1_amp_env.py:146-167): time_left -= step_dt; envs whose timer expired
//                    get command ~ U(lo, hi)^2 and time_left ~ U(t_lo, t_hi) -- and the reset-side resample of
//                    _reset_strategy_random (g1_amp_env.py:421-439): the same draw for the reset envs, or the fixed command
//                    (lo, 0) with an infinite timer when the range is empty.  The reference draws with torch.rand on the
//                    global CUDA generator after a nonzero() host sync; here the draw is counter-based (Philox4x32-10 keyed
//                    by seed, counter = (global env id, step), domain-separated from the reset-time draw of
//                    sample_times_kernel by the key's high word), so an env's command depends on (seed, step, env) only:
//                    no sync, no dependence on how the envs are sharded.  Parity with the reference is distributional by
//                    construction; bit-exact against oracle/rng.py::command_draw.
//   reward_log_means_kernel  the six to eight `.mean().item()` of _get_rewards (g1_amp_env.py:291-305) as ONE launch over
//                    reward_terms [T, N] -> means [T] on the device (fp64 accumulation, fixed order: deterministic); the
//                    host reads them only when extras["log"] is actually looked at.
#include "amp_common.hpp"
#include "command_kernels.hpp"

namespace amp {

__global__ __launch_bounds__(kBlock) void command_kernel(AmpCommandArgs a, int64_t N, int mode) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int64_t env = i;
  if (mode == AMP_COMMAND_RESET && a.env_ids) {
    int64_t n = a.n_ids;
    if (a.count) n = *a.count < n ? *a.count : n;
    if (i >= n) return;
    env = a.env_ids[i];
    if (env < 0 || env >= N) return;
  } else if (i >= N) {
    return;
  }
  const bool ranged = a.vel_span > 0.0f;
  if (mode == AMP_COMMAND_TICK) {
    const float left = a.time_left[env] - a.step_dt;  // command_time_left -= step_dt
    if (left <= 0.0f && ranged) {
      float cx, cy, tl;
      draw_command(a.seed, command_step_of(a), (uint64_t)(a.env_offset + env), 0u, a.vel_lo, a.vel_span, a.t_lo, a.t_span, cx, cy, tl);
      a.command[2 * env] = cx;
      a.command[2 * env + 1] = cy;
      a.time_left[env] = tl;
    } else {
      a.time_left[env] = left;
    }
    return;
  }
  if (!a.env_ids && !(a.reset_mask && a.reset_mask[env])) return;
  command_reset_env(a, env);
}

// The per-step bookkeeping in front of the physics step as ONE launch (amp_pre_physics_step): what G1AmpEnv._pre_physics_step
// (g1_amp_env.py:142-167) and _apply_action (:169-173) do with five ATen launches + the command-timer launch:
//   actions[e] = actions_in[e]; last_actions[e] = actions_in[e]; target[e] = offset[j] + scale[j] * actions_in[e]   (mul, add:
//   the reference's `self.action_offset + self.action_scale * self.actions`, bit for bit)   + the timers' tick for env < N.
// A pure streaming kernel (one 16-B load, up to three 16-B stores per lane: 30 MB at 65 536 envs): the first `copy_blocks`
// workgroups own four consecutive elements per lane (`vec`: every array 16-B aligned; else one element per lane), the
// workgroups behind them own one env per lane: `episode_length_buf += 1` and the command timer's tick -- the Philox draw of an
// expired timer diverges only there.
// (Round 3's body owned one element per lane and ticked from the joint-0 lanes: 10.7 us at 65 536 envs, 2.8 TB/s.)
__global__ __launch_bounds__(kBlock) void pre_physics_kernel(AmpPrePhysicsArgs a, AmpCommandArgs c, unsigned copy_blocks, int vec,
                                                             int has_tick) {
  typedef float pf4 __attribute__((ext_vector_type(4)));
  // the device-side step counter for the launches behind this one (the tick below draws with *step_in, a different word)
  if (a.step_out && blockIdx.x == 0 && threadIdx.x == 0) *a.step_out = *a.step_in + 1ull;
  if (blockIdx.x >= copy_blocks) {  // the per-env workgroups: episode-length increment, command timers
    const int64_t env = (int64_t)(blockIdx.x - copy_blocks) * kBlock + threadIdx.x;
    if (env >= a.num_envs) return;
    if (a.episode_length) a.episode_length[env] += 1;  // DirectRLEnv.step: self.episode_length_buf += 1
    if (!has_tick) return;
    const float left = c.time_left[env] - c.step_dt;  // command_time_left -= step_dt (command_kernel's TICK branch)
    if (left <= 0.0f && c.vel_span > 0.0f) {
      float cx, cy, tl;
      draw_command(c.seed, command_step_of(c), (uint64_t)(c.env_offset + env), 0u, c.vel_lo, c.vel_span, c.t_lo, c.t_span, cx, cy, tl);
      c.command[2 * env] = cx;
      c.command[2 * env + 1] = cy;
      c.time_left[env] = tl;
    } else {
      c.time_left[env] = left;
    }
    return;
  }
  const int64_t total = a.num_envs * a.n_actions;
  const int n = a.n_actions;
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  auto one = [&](const int64_t e, const int j, const float v) {
    if (a.actions) a.actions[e] = v;
    if (a.last_actions) a.last_actions[e] = v;
    if (a.target) {
      const float sv = a.scale ? a.scale[j] * v : v;
      a.target[e] = a.offset ? a.offset[j] + sv : sv;
    }
  };
  if (!vec) {
    if (q < total) one(q, (int)(q % n), a.actions_in[q]);
    return;
  }
  const int64_t e = 4 * q;
  if (e >= total) return;
  int j = (int)(e % n);
  if (e + 4 > total) {  // the last, partial quad
    for (int64_t i = e; i < total; ++i) {
      one(i, j, a.actions_in[i]);
      if (++j >= n) j = 0;
    }
    return;
  }
  const pf4 v = *reinterpret_cast<const pf4*>(a.actions_in + e);
  if (a.actions) *reinterpret_cast<pf4*>(a.actions + e) = v;
  if (a.last_actions) *reinterpret_cast<pf4*>(a.last_actions + e) = v;
  if (a.target) {
    pf4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sv = a.scale ? a.scale[j] * v[i] : v[i];
      t[i] = a.offset ? a.offset[j] + sv : sv;
      if (++j >= n) j = 0;
    }
    *reinterpret_cast<pf4*>(a.target + e) = t;
  }
}

// One workgroup (1 024 lanes) per term.  Lane t owns elements t*4 .. t*4+3 of every 4 096-element trip and accumulates them
// in fp64 in trip order; eight trips' 16-B loads are issued before the first add (the first version issued one dependent
// 4-B load per trip and 256 lanes: 64 us at 65 536 envs, all latency), then a fixed binary tree over the lanes.
constexpr int kMeansBlock = 1024;
__global__ __launch_bounds__(kMeansBlock) void reward_log_means_kernel(const float* __restrict__ terms, int64_t N,
                                                                       float* __restrict__ means) {
  typedef float mf4 __attribute__((ext_vector_type(4)));
  __shared__ double red[kMeansBlock];
  const float* row = terms + (int64_t)blockIdx.x * N;
  double s = 0.0;
  const int tid = threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
    const mf4* row4 = reinterpret_cast<const mf4*>(row);
    const int64_t n4 = N >> 2;
    for (int64_t i = tid; i < n4; i += 8 * kMeansBlock) {
      mf4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = i + k * kMeansBlock < n4 ? row4[i + k * kMeansBlock] : mf4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < 8; ++k) s += ((double)v[k][0] + (double)v[k][1]) + ((double)v[k][2] + (double)v[k][3]);
    }
    for (int64_t i = (n4 << 2) + tid; i < N; i += kMeansBlock) s += (double)row[i];
  } else {
    for (int64_t i = tid; i < N; i += kMeansBlock) s += (double)row[i];
  }
  red[tid] = s;
  __syncthreads();
  for (int o = kMeansBlock / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) means[blockIdx.x] = (float)(red[0] / (double)N);
}

}  // namespace amp

using namespace amp;

extern "C" {

int amp_command_step(const AmpCommandArgs* a, int64_t num_envs, int32_t mode, amp_stream_t stream) {
  AMP_REQUIRE(a, "amp_command_step: null argument");
  AMP_REQUIRE(num_envs >= 0, "amp_command_step: negative num_envs");
  AMP_REQUIRE(mode == AMP_COMMAND_TICK || mode == AMP_COMMAND_RESET, "amp_command_step: mode must be AMP_COMMAND_TICK (0) or AMP_COMMAND_RESET (1)");
  if (num_envs == 0) return AMP_OK;
  AMP_REQUIRE(a->command && a->time_left, "amp_command_step: null buffer");
  AMP_REQUIRE(mode == AMP_COMMAND_TICK || a->reset_mask || a->env_ids, "amp_command_step: the reset mode needs reset_mask or env_ids");
  AMP_REQUIRE(!(a->vel_span > 0.0f) || a->t_span >= 0.0f, "amp_command_step: negative resampling-time span");
  AMP_REQUIRE(a->env_ids == nullptr || a->n_ids >= 0, "amp_command_step: negative n_ids");
  const int64_t threads = (mode == AMP_COMMAND_RESET && a->env_ids) ? a->n_ids : num_envs;
  if (threads == 0) return AMP_OK;
  hipStream_t st = (hipStream_t)stream;
  { amp::TraceScope trace__("command_kernel", st);
    command_kernel<<<(unsigned)((threads + kBlock - 1) / kBlock), kBlock, 0, st>>>(*a, num_envs, mode);
  }
  return launch_status("command_kernel");
}

int amp_pre_physics_step(const AmpPrePhysicsArgs* a, const AmpCommandArgs* tick, amp_stream_t stream) {
  AMP_REQUIRE(a, "amp_pre_physics_step: null argument");
  AMP_REQUIRE(a->num_envs >= 0 && a->n_actions >= 1, "amp_pre_physics_step: need num_envs >= 0 and n_actions >= 1");
  if (a->num_envs == 0) return AMP_OK;
  AMP_REQUIRE(a->actions_in, "amp_pre_physics_step: actions_in is null");
  AMP_REQUIRE(!tick || (tick->command && tick->time_left), "amp_pre_physics_step: null command buffer");
  AMP_REQUIRE((a->step_in == nullptr) == (a->step_out == nullptr) && (!a->step_out || a->step_out != a->step_in),
              "amp_pre_physics_step: step_in / step_out come together and are different words");
  AMP_REQUIRE(!a->step_out || !tick || tick->step_dev != a->step_out, "amp_pre_physics_step: the tick must not read the word this launch writes");
  AMP_REQUIRE(!tick || !(tick->vel_span > 0.0f) || tick->t_span >= 0.0f, "amp_pre_physics_step: negative resampling-time span");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = a->num_envs * a->n_actions;
  const uintptr_t bits = reinterpret_cast<uintptr_t>(a->actions_in) | reinterpret_cast<uintptr_t>(a->actions) |
                         reinterpret_cast<uintptr_t>(a->last_actions) | reinterpret_cast<uintptr_t>(a->target);
  const int vec = (bits & 15) == 0;  // (null pointers are aligned)
  const int64_t lanes = vec ? (total + 3) / 4 : total;
  const unsigned copy_blocks = (unsigned)((lanes + kBlock - 1) / kBlock);
  const unsigned tick_blocks = (tick || a->episode_length) ? (unsigned)((a->num_envs + kBlock - 1) / kBlock) : 0u;
  { amp::TraceScope trace__("pre_physics_kernel", st);
    pre_physics_kernel<<<copy_blocks + tick_blocks, kBlock, 0, st>>>(*a, tick ? *tick : AmpCommandArgs{}, copy_blocks, vec, tick ? 1 : 0);
  }
  return launch_status("pre_physics_kernel");
}

int amp_reward_log_means(const float* reward_terms, int32_t n_terms, int64_t num_envs, float* means, amp_stream_t stream) {
  AMP_REQUIRE(reward_terms && means, "amp_reward_log_means: null buffer");
  AMP_REQUIRE(n_terms >= 1 && n_terms <= 64 && num_envs >= 1, "amp_reward_log_means: need 1..64 terms and >= 1 env");
  hipStream_t st = (hipStream_t)stream;
  { amp::TraceScope trace__("reward_log_means_kernel", st);
    reward_log_means_kernel<<<(unsigned)n_terms, kMeansBlock, 0, st>>>(reward_terms, num_envs, means);
  }
  return launch_status("reward_log_means_kernel");
}

}  // extern "C"
