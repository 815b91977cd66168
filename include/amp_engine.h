/*
 * amp_engine.h -- C ABI of libamp_engine.so: the MI355X (gfx950) AMP observation / motion-sample /
 * reward engine.  This is the drop-in boundary for the hot path of zhoushanghai/humanoid_amp
 * (SURVEY.md section 8b).  Every entry point cites the reference interface it replaces
 * (paths relative to the reference repository).
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch / C++ types.
 *   - every function returns int: AMP_OK (0) or a negative AMP_ERR_*; the message of the last failure
 *     on the calling thread is returned by amp_last_error().  No C++ exception crosses this boundary.
 *   - "dev" pointers are device (HBM) pointers owned by the CALLER; the library borrows them for the
 *     duration of the enqueue (or, where stated, for the life time of a handle) and never frees them.
 *   - all launches are asynchronous on the passed stream (hipStream_t, may be 0 = null stream).  No
 *     entry point synchronises the device, allocates or frees device memory after *_create, so every
 *     launch function is hipGraph-capturable.
 *   - quaternions are wxyz; all floating-point tables / outputs are fp32; times are fp64; ids int64.
 *   - handles are immutable after their *_create / *_set_* calls: re-entrant across streams.
 */
#ifndef AMP_ENGINE_H
#define AMP_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMP_ABI_VERSION 11

typedef void* amp_stream_t; /* hipStream_t */
typedef void* amp_event_t;  /* hipEvent_t  */
typedef struct AmpMotion AmpMotion;
typedef struct AmpDisc AmpDisc;

enum {
  AMP_OK = 0,
  AMP_ERR_INVALID = -1,     /* bad argument (null pointer, shape mismatch, out-of-range index) */
  AMP_ERR_HIP = -2,         /* a HIP runtime call failed; amp_last_error() has hipGetErrorString */
  AMP_ERR_UNSUPPORTED = -3, /* configuration outside what the kernels are built for */
  AMP_ERR_NO_DEVICE = -4    /* no gfx950 device visible */
};

int amp_abi_version(void);
const char* amp_last_error(void);
/* Name of the device the library would run on; fails with AMP_ERR_NO_DEVICE when there is none. */
int amp_device_name(char* buf, int64_t buf_len);

/* ------------------------------------------------------------------------------------------------
 * Kernel tracer (the reference has none: train.py:281,300 only wraps the whole run in time.time()).
 * While tracing is on, every kernel launch whose name contains `filter` (NULL / "" = all) is bracketed by a
 * pair of HIP events on its own stream.  Read the durations after synchronising those streams.
 * Record slots are handed out under a mutex (launches may come from several host threads); begin / end are meant
 * to be called from one controlling thread.
 * ------------------------------------------------------------------------------------------------ */
int amp_trace_begin(int64_t capacity, const char* filter);
/* Bracket only every `every`-th matching launch from now on (1 = all; reset to 1 by amp_trace_begin): an event pair
 * costs the queue a few microseconds, which a timed region of 300-us steps notices (bench.py samples its dominant kernel). */
int amp_trace_sample(int64_t every);
int amp_trace_end(void);                 /* stop recording (records stay readable until the next begin) */
int64_t amp_trace_count(void);
/* name and duration in milliseconds of record i; AMP_ERR_HIP if its events have not completed yet */
int amp_trace_get(int64_t i, char* name_buf, int64_t name_len, float* ms);

/* Measurement aid: one launch of a bare v_mfma_f32_32x32x16_f16 stream (one 4-wave workgroup per CU, 16 independent
 * accumulators, iters x 48 MFMAs per wave) on constant (random_operands = 0) or changing full-entropy operands (1).
 * *flops_out = FLOPs of the launch; time it with the tracer (kernel name "mfma_f16_calibration_kernel") or events.
 * scratch_dev: >= 256 floats per CU; with 4 more floats per CU the launch also leaves, behind those 256 * CUs floats, two
 * uint64 per workgroup: shader-clock ticks (s_memtime) and 100 MHz wall ticks (s_memrealtime) across the MFMA loop --
 * their ratio is the core clock the chip sustained under the load.  What the result is for: MI355X is power-limited, the matrix pipes sustain 0.55-0.62
 * of the nominal fp16 peak on changing operands -- the ceiling the discriminator GEMMs' roofline fraction is read against.
 * random_operands bit 1 (values 2 / 3): the stream of layer 2 since round 3 instead -- v_mfma_f32_16x16x32_f16, one 8-wave
 * workgroup per CU (two waves per SIMD), iters x 96 MFMAs per wave, 512 floats (+ 4) of scratch per CU; on changing operands it
 * sustains ~13 % more than the 32 x 32 x 16 stream. */
int amp_calibrate_mfma_f16(int32_t random_operands, int32_t iters, float* scratch_dev, int64_t scratch_floats,
                           double* flops_out, amp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Motion table  (replaces MotionLoader.__init__, motions/motion_loader.py:98-164)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t n_clips;
  int32_t n_dof;
  int32_t n_bodies;
  int32_t reserved;
  int64_t n_frames;           /* total frames of the concatenated tables */
  double dt;                  /* 1/fps of the FIRST clip (motion_loader.py:122) */
  const int64_t* clip_frames; /* HOST [n_clips] frames per clip, each >= 2 */
  /* device tables, row-major fp32, BORROWED for the life time of the handle */
  const float* dof_positions;           /* dev [F, n_dof]       */
  const float* dof_velocities;          /* dev [F, n_dof]       */
  const float* body_positions;          /* dev [F, n_bodies, 3] */
  const float* body_rotations;          /* dev [F, n_bodies, 4] wxyz */
  const float* body_linear_velocities;  /* dev [F, n_bodies, 3] */
  const float* body_angular_velocities; /* dev [F, n_bodies, 3] */
} AmpMotionDesc;

int amp_motion_create(const AmpMotionDesc* desc, AmpMotion** out);
int amp_motion_destroy(AmpMotion* h);

/* Select the AMP hot subset: robot-order DoF permutation, reference body and key bodies
 * (replaces the index plumbing of g1_amp_env.py:47-60,478-484).  Builds the handle's private
 * [F, 2*n_dof + 13 + 3*n_key] device table.  dof_perm / key_bodies are HOST arrays. */
int amp_motion_set_obs_layout(AmpMotion* h, const int32_t* dof_perm, int32_t ref_body,
                              const int32_t* key_bodies, int32_t n_key, amp_stream_t stream);

/* (t, clip) -> bracketing frame indices + signed blend; fp64/int64 numpy semantics.
 * Replaces MotionLoader._compute_frame_blend (motion_loader.py:281-307).
 * motion_ids may be NULL (clip 0 for every sample, motion_loader.py:365-366). */
int amp_motion_frame_blend(const AmpMotion* h, const double* times_dev, const int64_t* motion_ids_dev, int64_t n,
                           int64_t* index_0_dev, int64_t* index_1_dev, double* blend_dev, amp_stream_t stream);

/* Replaces MotionLoader.sample for explicit times (motion_loader.py:331-390): 5x LERP + body SLERP.
 * Outputs [n,n_dof] x2, [n,B,3], [n,B,4], [n,B,3], [n,B,3]; any output pointer may be NULL (skipped). */
int amp_motion_sample(const AmpMotion* h, const double* times_dev, const int64_t* motion_ids_dev, int64_t n,
                      float* dof_pos_dev, float* dof_vel_dev, float* body_pos_dev, float* body_rot_dev,
                      float* body_lin_vel_dev, float* body_ang_vel_dev, amp_stream_t stream);

/* Expert AMP observations: K history frames t - dt*k (newest first) per sample, fused with the feature
 * extraction.  Replaces G1AmpEnv.collect_reference_motions (g1_amp_env.py:445-486) + compute_obs (:535-561).
 * out_dev is [*, K, D] with D = 2*n_dof + 13 + 3*n_key.  Row r is written to row dst_rows_dev[r] when
 * dst_rows_dev != NULL (the reset scatter amp_observation_buffer[env_ids] = ..., g1_amp_env.py:417), else r. */
int amp_collect_reference(const AmpMotion* h, const double* times_dev, const int64_t* motion_ids_dev, int64_t n,
                          int32_t K, float* out_dev, const int64_t* dst_rows_dev, amp_stream_t stream);

/* Reference-state initialisation of reset envs: root_state [n,13] = (pos + env_origins[env_ids], z += z_lift,
 * quat wxyz, lin vel, ang vel) of the reference body, DoF pos/vel in robot order.
 * Replaces the sampling half of G1AmpEnv._reset_strategy_random (g1_amp_env.py:385-411).
 * env_origins_dev is [num_envs,3]; env_ids_dev [n] indexes it (NULL: row r). */
int amp_reset_reference_state(const AmpMotion* h, const double* times_dev, const int64_t* motion_ids_dev,
                              const int64_t* env_ids_dev, int64_t n, const float* env_origins_dev, float z_lift,
                              float* root_state_dev, float* dof_pos_dev, float* dof_vel_dev, amp_stream_t stream);

/* Device-side MotionLoader.sample_times (motion_loader.py:309-329) for the reset path: counter-based Philox4x32-10,
 * key = seed, counter = (index_dev[i] or i, step).  clip ~ U{0..n_clips-1}, t = 0 (start != 0) or U[0,1) * duration.
 * n_dev (device int64, may be NULL) caps the number of draws at min(n, *n_dev) without a host read-back.
 * Parity with the reference's host numpy RNG is distributional only; bit-exact against oracle/rng.py. */
int amp_motion_sample_times(const AmpMotion* h, uint64_t seed, uint64_t step, int32_t start, const int64_t* index_dev,
                            const int64_t* n_dev, int64_t n, int64_t* motion_ids_dev, double* times_dev, amp_stream_t stream);

/* The whole reference-state reset of G1AmpEnv._reset_strategy_random (g1_amp_env.py:371-419) driven by the DEVICE-side
 * output of amp_reset_compact* -- no count read-back, no host RNG: for i < *count, env = env_ids[i]:
 * (clip, t) drawn with (seed, step, env_offset + env); root_state / dof rows i; amp_obs_buffer[env] = K expert frames. */
typedef struct {
  const int64_t* env_ids;   /* dev [max_n] ascending reset ids */
  const int64_t* count;     /* dev [1] */
  int64_t max_n;            /* capacity of the compact outputs (num_envs) */
  uint64_t seed, step;
  int32_t start;            /* reset_strategy "random-start" */
  int32_t K;
  const float* env_origins; /* dev [num_envs, 3] or NULL */
  float z_lift;
  int32_t mode;             /* AMP_RESET_REFERENCE (0) or AMP_RESET_DEFAULT (1), see below */
  float* root_state;        /* dev [max_n, 13] compact, may be NULL */
  float* dof_pos;           /* dev [max_n, n_dof] compact, may be NULL */
  float* dof_vel;
  float* amp_obs_buffer;    /* dev [num_envs, K, D], rows env_ids[i] overwritten; may be NULL */
  int64_t* motion_ids;      /* dev [max_n] compact out */
  double* motion_times;     /* dev [max_n] compact out */
  /* optional per-env mirrors of the draw, rows env_ids[i] (the env attributes g1_amp_env.py:377-382 keeps) */
  int64_t* env_motion_ids;      /* dev [num_envs] or NULL */
  float* env_motion_start_times; /* dev [num_envs] fp32 or NULL */
  int64_t env_offset;           /* global id of env 0 of this shard: draws are keyed by env_offset + env_ids[i] */
  /* optional per-env clears of the reset envs (DirectRLEnv._reset_idx: episode_length_buf[env_ids] = 0;
   * g1_amp_env.py:352-358: last_actions[env_ids] = 0, _just_reset_mask[env_ids] = True) */
  int64_t* episode_length;      /* dev [num_envs] or NULL */
  float* last_actions;          /* dev [num_envs, n_actions] or NULL */
  uint8_t* just_reset;          /* dev [num_envs] or NULL */
  int32_t n_actions;
  int32_t reserved2;
  /* optional device-side step counter: the draws use step + *step_dev.  Lets a captured hipGraph of the env step advance
   * the counter-based streams from one replay to the next (the caller increments *step_dev inside the graph). */
  const uint64_t* step_dev;
  /* mode == AMP_RESET_DEFAULT -- reset_strategy "default" (g1_amp_env.py:338-339, 362-369): rows i < *count of
   * root_state / dof_pos / dof_vel get default_root_state[env] (+ env_origins[env] on x, y, z) / default_joint_pos[env] /
   * default_joint_vel[env]; the per-env clears run; NOTHING else is touched (no clip / time draw, amp_obs_buffer and the
   * command stay as they are, exactly as in the reference).  All three are required in that mode, ignored otherwise. */
  const float* default_root_state;  /* dev [num_envs, 13] */
  const float* default_joint_pos;   /* dev [num_envs, n_dof] */
  const float* default_joint_vel;   /* dev [num_envs, n_dof] */
  /* optional (amp_reset_compact_apply only; needs step_dev): *step_dev_out = *step_dev, written by one thread -- the other half
   * of AmpPrePhysicsArgs.step_in / step_out.  Must not alias step_dev. */
  uint64_t* step_dev_out;
} AmpResetArgs;
#define AMP_RESET_REFERENCE 0
#define AMP_RESET_DEFAULT 1
int amp_reset_apply(const AmpMotion* h, const AmpResetArgs* args, amp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Per-step env kernels  (replace G1AmpEnv._get_dones / _get_rewards / _get_observations,
 * g1_amp_env.py:175-242,246-330 and humanoid_amp_env.py:105-140)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t n_dof;
  int32_t n_key;                /* key bodies (4) */
  int32_t num_amp_observations; /* K */
  int32_t num_actor_observations; /* stacked actor frames (>= 1), g1_amp_env_cfg.py:47,176 */
  int32_t use_last_actions;     /* 1: G1 policy obs = base | last_actions [| command]; 0: humanoid (policy obs = AMP frame) */
  int32_t use_command;          /* rew_track_vel > 0 */
  int32_t history_include_last_actions;
  int32_t history_include_command;
  int32_t early_termination;
  int32_t reward_mode;          /* 0: constant 1 (humanoid_amp_env.py:128-129); 1: G1 task reward */
  int64_t max_episode_length;
  float termination_height;
  float rew_termination, rew_action_l2, rew_joint_pos_limits, rew_joint_acc_l2, rew_joint_vel_l2;
  /* fp64 on purpose: the reference derives threshold / slope from python floats before rounding to fp32 */
  double rew_track_vel;         /* weight of the velocity-tracking reward (g1_amp_env.py:262) */
  double track_sigma;           /* 0.5  (g1_amp_env.py:263) */
  double track_floor;           /* 4.0  (g1_amp_env.py:264) */
} AmpEnvCfg;

/* Simulator state, strided views (strides in ELEMENTS between consecutive envs) so Isaac Lab's AoS
 * tensors (robot.data.body_pos_w[:, ref] ...) can be passed without a gather. */
typedef struct {
  const float* joint_pos;    int64_t joint_pos_stride;    /* [N, n_dof] */
  const float* joint_vel;    int64_t joint_vel_stride;    /* [N, n_dof] */
  const float* joint_acc;    int64_t joint_acc_stride;    /* [N, n_dof]  (reward only) */
  const float* actions;      int64_t actions_stride;      /* [N, n_dof]  (reward only) */
  const float* root_pos;     int64_t root_pos_stride;     /* [N, 3] reference body */
  const float* root_quat;    int64_t root_quat_stride;    /* [N, 4] wxyz */
  const float* root_lin_vel; int64_t root_lin_vel_stride; /* [N, 3] world */
  const float* root_ang_vel; int64_t root_ang_vel_stride; /* [N, 3] world */
  const float* body_pos;     int64_t body_pos_stride;     /* [N, n_bodies_robot, 3] */
  int32_t key_body[8];                                    /* robot body indices of the key bodies */
  const float* soft_limits;  int64_t soft_limits_stride;  /* [N, n_dof, 2]; stride 0 = one row for all envs */
  const int64_t* episode_length;                          /* [N] */
  const float* command;                                   /* [N, 2] */
  const float* last_actions;                              /* [N, n_dof] */
} AmpSimState;

typedef struct {
  float* amp_obs_buffer;     /* [N, K, D] in/out, slot 0 = newest (g1_amp_env.py:67-74,187-190) */
  float* policy_obs;         /* [N, P] */
  float* actor_history;      /* [N, n_actor-1, per_frame] in/out, NULL when n_actor == 1 */
  uint8_t* just_reset;       /* [N] in/out bool, NULL when n_actor == 1 (g1_amp_env.py:117-119,216-222) */
  float* reward;             /* [N] task reward */
  float* reward_terms;       /* optional [8, N]: total, track, track_err, termination, action_l2, limits, acc_l2, vel_l2 */
  uint8_t* died;             /* [N] bool */
  uint8_t* time_out;         /* [N] bool */
  uint8_t* reset_mask;       /* [N] died | time_out */
  int32_t* reset_tile_counts; /* optional [ceil(N / amp_env_step_tile_envs(cfg, N))]: reset envs per tile (feeds amp_reset_compact_tiles) */
  /* Optional fusion of the discriminator's input scaler into the OBS phase (saves one pass over amp_obs):
   * disc_input receives v = clamp((amp_obs - mean) / den, -clip, clip) for the K*D columns in the layout
   * amp_disc_input_layout() reports (every field below comes from it; scaler_mean == NULL copies unscaled):
   *   AMP_DISC_INPUT_F32_ROWS   float       [N, disc_input_stride]   <- v
   *   AMP_DISC_INPUT_F16_BLOCKS _Float16 [N, disc_input_stride / 32, 2, 32]: per row and k-block of 32 columns the plane
   *                             p0 = rn16(s v) of the 32 columns, then the plane p1 = rn16(s v - p0), s = disc_plane_scale
   *                             -- the 128-B "block layout" the GEMM kernels fill their LDS stages from
   * (both 4 bytes per element).  The padding columns are never written: zero the buffer once.  Feed it to
   * amp_disc_style_reward_prescaled(). */
  void* disc_input;
  int64_t disc_input_stride;
  const float* scaler_mean;
  const float* scaler_den;
  float scaler_clip;
  int32_t disc_input_format;
  float disc_plane_scale;
  int32_t amp_obs_read_next; /* 1: the kernel that follows reads amp_obs_buffer (a discriminator taking the raw rows): store it with the
                              * default cache policy; 0: non-temporal stores (nothing in the step reads the rows back) */
} AmpEnvBuffers;

enum { AMP_PHASE_DONES = 1, AMP_PHASE_REWARD = 2, AMP_PHASE_OBS = 4 };

/* amp_env_step + amp_collect_reference (dst_rows = NULL) as ONE launch: some workgroups run the env step, the others
 * the expert-motion sample for (times_dev, motion_ids_dev) [n_samples] -> expert_out_dev [n_samples, K * D].  The two
 * have no data dependence; sharing a launch lets them overlap.  Results are bit-identical to the separate calls. */
int amp_env_step_with_reference(const AmpEnvCfg* cfg, const AmpSimState* state, const AmpEnvBuffers* bufs, int64_t num_envs,
                                uint32_t phases, const AmpMotion* motion, const double* times_dev,
                                const int64_t* motion_ids_dev, int64_t n_samples, int32_t K, float* expert_out_dev,
                                amp_stream_t stream);

/* Envs per workgroup tile amp_env_step uses for this configuration and a shard of num_envs (8, 16 or 32; -1 for an
 * invalid cfg): the granularity of reset_tile_counts. */
int32_t amp_env_step_tile_envs(const AmpEnvCfg* cfg, int64_t num_envs);
/* Size of one policy observation row for a configuration (g1_amp_env_cfg.py:186-206). */
int64_t amp_policy_obs_size(const AmpEnvCfg* cfg);
/* Size of one actor-history frame (g1_amp_env.py:96-107); 0 when num_actor_observations == 1. */
int64_t amp_actor_history_frame_size(const AmpEnvCfg* cfg);

/* One pass over the N envs doing the requested phases (any OR of AMP_PHASE_*), in the reference's
 * order dones -> reward -> observations.  Pointers a phase does not need may be NULL. */
int amp_env_step(const AmpEnvCfg* cfg, const AmpSimState* state, const AmpEnvBuffers* bufs, int64_t num_envs,
                 uint32_t phases, amp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Velocity-command timers (replaces the body of G1AmpEnv._pre_physics_step, g1_amp_env.py:146-167, and the command
 * resample of _reset_strategy_random, g1_amp_env.py:421-439; ranges g1_amp_env_cfg.py:96-100).
 *   AMP_COMMAND_TICK   every env: time_left -= step_dt; where it drops to <= 0 and vel_span > 0:
 *                      command ~ U(vel_lo, vel_lo + vel_span)^2, time_left ~ U(t_lo, t_lo + t_span).
 *   AMP_COMMAND_RESET  the envs of reset_mask (any non-zero byte), or env_ids[i] for i < min(n_ids, *count): the same
 *                      draw; with vel_span <= 0 the fixed command (vel_lo, 0) and an infinite timer.
 * The reference draws with torch.rand behind a nonzero() host sync; here every draw is Philox4x32-10 keyed by
 * (seed, step, env_offset + env) -- no sync, and the command of an env does not depend on how envs are sharded.
 * Parity with the reference is distributional only; bit-exact against oracle/rng.py.
 * ------------------------------------------------------------------------------------------------ */
enum { AMP_COMMAND_TICK = 0, AMP_COMMAND_RESET = 1 };
typedef struct {
  float* command;            /* dev [N, 2] in/out  (command_target_speed) */
  float* time_left;          /* dev [N] in/out     (command_time_left) */
  float step_dt;             /* sim.dt * decimation */
  float vel_lo, vel_span;    /* track_vel_range[0], (float)(track_vel_range[1] - track_vel_range[0]) */
  float t_lo, t_span;        /* command_resampling_time_range likewise */
  int32_t reserved;
  uint64_t seed, step;
  int64_t env_offset;        /* global id of env 0 of this shard */
  const uint8_t* reset_mask; /* RESET: dev [N], or NULL when env_ids is given */
  const int64_t* env_ids;    /* RESET: dev [n_ids] local env ids, or NULL */
  const int64_t* count;      /* RESET: dev [1] optional cap on n_ids (the compaction's count), or NULL */
  int64_t n_ids;
  const uint64_t* step_dev;  /* optional device-side step counter: the draws use step + *step_dev (see AmpResetArgs) */
} AmpCommandArgs;
int amp_command_step(const AmpCommandArgs* args, int64_t num_envs, int32_t mode, amp_stream_t stream);

/* The bookkeeping in front of the physics step as ONE launch: G1AmpEnv._pre_physics_step (g1_amp_env.py:142-167) and the
 * arithmetic of _apply_action (:169-173) -- `self.actions = actions.clone()`, the command timers' tick (tick != NULL: exactly
 * amp_command_step(AMP_COMMAND_TICK)), `target = action_offset + action_scale * actions` (one fp32 multiply, one add, as the
 * reference evaluates it), `self.last_actions = self.actions.clone()` -- instead of five ATen launches and the timer launch.
 * Any output pointer may be NULL (skipped); offset / scale NULL mean 0 / 1. */
typedef struct {
  const float* actions_in;   /* dev [num_envs, n_actions] the agent's actions */
  float* actions;            /* dev [num_envs, n_actions] out: self.actions */
  float* last_actions;       /* dev [num_envs, n_actions] out: self.last_actions */
  float* target;             /* dev [num_envs, n_actions] out: joint position targets */
  const float* offset;       /* dev [n_actions] action_offset, or NULL */
  const float* scale;        /* dev [n_actions] action_scale, or NULL */
  int64_t num_envs;
  int32_t n_actions;
  int32_t reserved;
  /* optional: DirectRLEnv.step's `self.episode_length_buf += 1` (Isaac Lab does it between the physics step and _get_dones;
   * nothing reads the buffer in between, so it may ride on this launch: one ATen launch fewer per env step) */
  int64_t* episode_length;   /* dev [num_envs] int64, or NULL */
  /* optional: the device-side step counter's hand-over inside a captured env step (DirectRLEnv.step's
   * `common_step_counter += 1` sits between this launch and the reset): *step_out = *step_in + 1, written by one thread.  The
   * tick of THIS launch draws with *step_in (its AmpCommandArgs.step_dev); the reset launch reads *step_out as its step_dev and
   * copies it back (AmpResetArgs.step_dev_out = step_in), so the next step starts one further -- no increment launch of its
   * own in the graph.  step_in and step_out must be different words; both NULL = off. */
  const uint64_t* step_in;
  uint64_t* step_out;
} AmpPrePhysicsArgs;
int amp_pre_physics_step(const AmpPrePhysicsArgs* args, const AmpCommandArgs* tick, amp_stream_t stream);

/* Means over the envs of the rows of reward_terms [n_terms, N] (AmpEnvBuffers.reward_terms) -> means_dev [n_terms]:
 * the `.mean().item()` chain of G1AmpEnv._get_rewards (g1_amp_env.py:291-305) as one launch with no read-back (fp64
 * accumulation in a fixed order). */
int amp_reward_log_means(const float* reward_terms_dev, int32_t n_terms, int64_t num_envs, float* means_dev,
                         amp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Reset-index compaction  (replaces reset_buf.nonzero(as_tuple=False).squeeze(-1) of DirectRLEnv.step,
 * consumed by G1AmpEnv._reset_idx, g1_amp_env.py:332-358).  Ascending int64 ids, bit-exact.
 * ------------------------------------------------------------------------------------------------ */
int64_t amp_reset_compact_workspace_bytes(int64_t num_envs);
/* mask_dev [N] (any non-zero byte = reset) -> ids_dev [<= N] ascending, count_dev [1] int64. */
int amp_reset_compact(const uint8_t* mask_dev, int64_t num_envs, int64_t* ids_dev, int64_t* count_dev,
                      void* workspace_dev, amp_stream_t stream);
/* Same, re-using the per-tile counts amp_env_step(AMP_PHASE_DONES) already produced (tile_envs =
 * amp_env_step_tile_envs(cfg, num_envs): 8, 16, 32 or 64). */
int amp_reset_compact_tiles(const uint8_t* mask_dev, const int32_t* tile_counts_dev, int32_t tile_envs, int64_t num_envs,
                            int64_t* ids_dev, int64_t* count_dev, amp_stream_t stream);

/* Arguments of the compaction when it rides on another launch (amp_reset_compact_apply, the fused step tail). */
typedef struct {
  const uint8_t* mask;         /* dev [num_envs] reset mask (AmpEnvBuffers.reset_mask) */
  const int32_t* tile_counts;  /* dev: AmpEnvBuffers.reset_tile_counts of the same step */
  int32_t tile_envs;           /* amp_env_step_tile_envs(cfg, num_envs) */
  int32_t reserved;
  int64_t num_envs;
  int64_t* ids;                /* dev [num_envs] out, ascending */
  int64_t* count;              /* dev [1] out */
} AmpCompactArgs;

/* Device-count-bounded row scatter: the `tensor[env_ids] = rows` of the reset path (the write_*_to_sim calls and the
 * per-env clears of G1AmpEnv._reset_idx, g1_amp_env.py:348-358) for a state provider that holds plain device arrays, when
 * the number of ids lives on the device (amp_reset_compact*'s count): no read-back.  For every op, i < min(max_n, *count),
 * r < repeat, c < width:   dst[ids[i] * dst_stride + r * width + c] = (src ? src[i * src_stride + c] : fill) + (add ? add[r * width + c] : 0).
 * Up to 8 ops share one launch. */
typedef struct {
  const float* src;      /* dev [max_n, src_stride] compact rows, or NULL: write `fill` */
  int64_t src_stride;    /* elements between consecutive src rows */
  float fill;
  int32_t width;         /* floats copied per (row, repeat) */
  int32_t repeat;        /* the same `width` source floats written `repeat` times, back to back (e.g. once per body) */
  int32_t reserved;
  const float* add;      /* dev [repeat, width] added to every destination row, or NULL */
  float* dst;            /* dev [*, dst_stride] */
  int64_t dst_stride;    /* elements between consecutive dst rows (>= width * repeat) */
} AmpScatterRows;
int amp_scatter_rows(const AmpScatterRows* ops, int32_t n_ops, const int64_t* ids_dev, const int64_t* count_dev, int64_t max_n,
                     amp_stream_t stream);

/* amp_reset_compact_tiles + amp_reset_apply (+ the reset-side amp_command_step(AMP_COMMAND_RESET) when command != NULL) as
 * ONE launch: what DirectRLEnv.step does between _get_rewards and _get_observations (reset_buf.nonzero() -> _reset_idx,
 * g1_amp_env.py:332-441) with no host round trip and a single kernel.  args->env_ids / args->count must be compact->ids /
 * compact->count (they are written by this call), args->max_n >= compact->num_envs; args->amp_obs_buffer is required.
 * Results are bit-identical to the separate calls. */
/* log != NULL: the step's reward-log means (amp_reward_log_means of log->reward_terms [n_terms, num_envs] -> log->means
 * [n_terms]) ride on the same launch, one extra workgroup per term: one launch fewer per env step. */
typedef struct {
  const float* reward_terms;   /* dev [n_terms, num_envs] (AmpEnvBuffers.reward_terms) */
  int32_t n_terms;
  int32_t reserved;
  float* means;                /* dev [n_terms] out */
} AmpRewardLogArgs;
int amp_reset_compact_apply(const AmpMotion* h, const AmpCompactArgs* compact, const AmpResetArgs* args,
                            const AmpCommandArgs* command, const AmpRewardLogArgs* log, amp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Discriminator style reward  (replaces the inference half of skrl's AMP agent: amp_state_preprocessor
 * -> discriminator MLP -> style reward -> reward mix; shape agents/skrl_g1_walk_amp_cfg.yaml:31-39,
 * scales :88-95, scaler :77-78.  skrl itself is third-party: parity unpinned)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t in_dim;  /* K * D */
  int32_t h1;      /* 1024 */
  int32_t h2;      /* 512  */
  int32_t reserved;
  /* device, torch.nn.Linear layout: weight [out, in] row-major, bias [out]; COPIED at create */
  const float* w1; const float* b1;
  const float* w2; const float* b2;
  const float* w3; const float* b3;
} AmpDiscDesc;

int amp_disc_create(const AmpDiscDesc* desc, amp_stream_t stream, AmpDisc** out);
int amp_disc_destroy(AmpDisc* h);
/* load_state_dict-style update (skrl checkpoints, play.py:205-208): device-to-device copies of new weights of the SAME
 * shape into the handle's own buffers + refresh of the derived fp16 planes, asynchronous on `stream`, no allocation
 * (capturable).  The handle, its scaler, its precision mode and every device pointer it handed out
 * (amp_disc_input_layout, an attached trainer) stay valid.  The source buffers are read on `stream` only. */
int amp_disc_set_weights(AmpDisc* h, const AmpDiscDesc* desc, amp_stream_t stream);
/* RunningStandardScaler statistics (fp64 on device, as skrl keeps them); NULL mean disables scaling. */
int amp_disc_set_scaler(AmpDisc* h, const double* running_mean_dev, const double* running_variance_dev,
                        float epsilon, float clip_threshold, amp_stream_t stream);
/* GEMM engine of the forward pass.  Both deliver fp32-class accuracy (the path's 1e-5 budget; measured <= 1e-6 on O(1)
 * logits): AMP_DISC_F16X3 (default) carries every fp32 operand as two fp16 planes and issues three fp16 MFMAs per
 * k-step into one fp32 accumulator; AMP_DISC_FP32 runs the fp32 MFMA on fp32 operands (exact fma chain, ~2.7x slower). */
enum { AMP_DISC_F16X3 = 0, AMP_DISC_FP32 = 1 };
int amp_disc_set_precision(AmpDisc* h, int32_t mode, amp_stream_t stream);
int64_t amp_disc_workspace_bytes(const AmpDisc* h, int64_t rows);
/* Layout of the scaled input the GEMMs consume, for producers that write it directly (amp_env_step's fused scaler).
 * AMP_DISC_F16X3 with a clamping scaler consumes the two fp16 planes in block layout (the clamp bounds the plane
 * scale); every other configuration consumes fp32 rows.  Device pointers stay valid until the next
 * amp_disc_set_scaler / destroy. */
enum { AMP_DISC_INPUT_F32_ROWS = 0, AMP_DISC_INPUT_F16_BLOCKS = 1 };
typedef struct {
  int32_t format;          /* AMP_DISC_INPUT_* */
  int32_t padded_dim;      /* row length in elements: K*D zero-padded to the layer-1 k-tile; rows 16-B aligned */
  const float* mean_dev;   /* fp32 scaler vectors; NULL mean when no scaler is set */
  const float* den_dev;
  float clip;
  float plane_scale;       /* F16_BLOCKS: the power of two s with s * clip < 2^15 */
} AmpDiscInputLayout;
int amp_disc_input_layout(const AmpDisc* h, AmpDiscInputLayout* out);
/* Which kernels a style-reward call of `rows` rows would launch on this handle, and which process-environment switches are in
 * force (they are read once, at the first call that needs them: AMP_DISC_FUSED=0 switches the fused two-layer plan off,
 * AMP_DISC_FUSED_MIN_ROWS moves its threshold, AMP_TRAIN_FORK=0 / AMP_TRAIN_BK32=0 / AMP_TRAIN_F16_BIG=1 change the training step's
 * schedule / k-tile / fp16-pipe tile:
 * A/B switches for measurements, never needed for correctness -- every plan gives the same bits per row).  Pure host
 * arithmetic; a run can log the plan it took.  (Replaces nothing in the reference: skrl's discriminator forward is one
 * torch.nn.Sequential call, agents/skrl_g1_walk_amp_cfg.yaml:31-39.) */
enum { AMP_DISC_PLAN_REGISTER = 0, AMP_DISC_PLAN_DMA_128 = 1, AMP_DISC_PLAN_DMA_256_128 = 2, AMP_DISC_PLAN_DMA_256 = 3,
       AMP_DISC_PLAN_DMA_128_64 = 4, AMP_DISC_PLAN_FP32 = 16 };
enum { AMP_ENV_DISC_FUSED = 1, AMP_ENV_DISC_FUSED_MIN_ROWS = 2, AMP_ENV_TRAIN_FORK = 4, AMP_ENV_TRAIN_BK32 = 8, AMP_ENV_TRAIN_F16_BIG = 16 };
typedef struct {
  int32_t precision;        /* AMP_DISC_F16X3 / AMP_DISC_FP32 */
  int32_t plan;             /* AMP_DISC_PLAN_*: tiles of the two-kernel (layer 1, layer 2) launches that take the rows NOT on the fused kernel */
  int64_t fused_rows;       /* leading rows that run both layers as ONE launch (disc_mlp_fused_kernel); 0: none */
  int64_t chunk_rows;       /* rows per (layer 1, layer 2) launch pair of the remaining rows */
  int64_t fused_min_rows;   /* the fused plan's threshold in force (INT64_MAX: switched off) */
  int32_t env_overrides;    /* AMP_ENV_* bits: which of the environment switches are SET in this process */
  int32_t cu_count;         /* compute units the plans were sized for */
  int32_t raw_input;        /* 1: the whole batch takes the fused kernel AND that kernel can read fp32 observation rows itself (clamping
                             * scaler, even in_dim): amp_disc_style_reward(_compact) then run no scaler pass, and an env step need not
                             * produce disc_input for this batch size (8-B aligned rows of even stride assumed) */
  int32_t reserved;
} AmpDiscPlanInfo;
int amp_disc_plan_info(const AmpDisc* h, int64_t rows, AmpDiscPlanInfo* out);
/* Per-handle plan choice, overriding the process environment and the measured defaults (what tests and A/B measurements use
 * instead of environment variables): fused = -1 automatic / 0 never / 1 allowed with the threshold `fused_min_rows` (-1: the default
 * 24 576 rows).  Host-side state of the handle: call it between launches, not concurrently with them.  Every plan scores a row
 * the same bit for bit. */
int amp_disc_set_plan(AmpDisc* h, int32_t fused, int64_t fused_min_rows);
/* amp_obs_dev [rows, in_dim] (row stride in elements) -> logits [rows], style [rows] =
 * -log(max(1 - sigmoid(logit), 1e-4)) * reward_scale, combined = task_w * task + style_w * style.
 * logits / style / task / combined may be NULL.
 * amp_obs and task_reward are read only by the first kernel (scaler pass, which also snapshots the task reward
 * into the workspace); `inputs_consumed` (hipEvent_t, may be NULL) is recorded on `stream` right after it, so a
 * caller running the env on another stream may overwrite both buffers (the in-place AMP history shift of the next
 * env step) as soon as that event has completed, while the GEMMs are still running.  Exception: where the whole batch takes the fused
 * two-layer kernel straight from the fp32 rows (amp_disc_plan_info: raw_input; no scaler pass exists then) the rows and the task reward
 * are read until the call's last kernel, and the event is recorded behind it. */
int amp_disc_style_reward(const AmpDisc* h, const float* amp_obs_dev, int64_t rows, int64_t row_stride,
                          float reward_scale, const float* task_reward_dev, float task_weight, float style_weight,
                          float* logits_dev, float* style_dev, float* combined_dev, void* workspace_dev,
                          amp_event_t inputs_consumed, amp_stream_t stream);

/* Same as amp_disc_style_reward for an input that is already scaled, padded and in the layout amp_disc_input_layout()
 * reports (amp_env_step's disc_input; fp32 rows or fp16 plane blocks, padding columns zero): skips the
 * scaler pass. */
int amp_disc_style_reward_prescaled(const AmpDisc* h, const void* scaled_dev, int64_t rows, float reward_scale,
                                    const float* task_reward_dev, float task_weight, float style_weight,
                                    float* logits_dev, float* style_dev, float* combined_dev, void* workspace_dev,
                                    amp_stream_t stream);

/* amp_disc_style_reward_prescaled + amp_reset_compact_tiles with ONE tail launch: the reset-id compaction (independent of
 * the GEMMs) rides on the workgroups of the finalize launch instead of paying its own ~6 us launch -- what matters on the
 * 8 192-env shards of the multi-GPU configurations, where the step is a chain of latency-bound launches.  Results are
 * bit-identical to the two separate calls; ids / count become valid when this call's work completes. */
int amp_disc_style_reward_prescaled_compact(const AmpDisc* h, const void* scaled_dev, int64_t rows, float reward_scale,
                                            const float* task_reward_dev, float task_weight, float style_weight,
                                            float* logits_dev, float* style_dev, float* combined_dev, void* workspace_dev,
                                            const AmpCompactArgs* compact, amp_stream_t stream);

/* amp_disc_style_reward + amp_reset_compact_tiles with ONE tail launch, on the RAW observation rows: where the whole batch takes the
 * one-launch two-layer kernel (amp_disc_plan_info: fused_rows == rows; fp16-split engine, clamping scaler, even in_dim / row_stride,
 * 8-B aligned rows) that kernel reads the fp32 rows itself and applies scaler, clamp and plane split to the 48 elements a lane
 * holds -- no scaler pass, no scaled copy of the input in memory; everywhere else it is the two calls back to back.  Same results
 * bit for bit as amp_disc_style_reward (skrl's amp_state_preprocessor + discriminator forward, third-party: parity unpinned;
 * agents/skrl_g1_walk_amp_cfg.yaml:31-39,77-78). */
int amp_disc_style_reward_compact(const AmpDisc* h, const float* amp_obs_dev, int64_t rows, int64_t row_stride, float reward_scale,
                                  const float* task_reward_dev, float task_weight, float style_weight, float* logits_dev,
                                  float* style_dev, float* combined_dev, void* workspace_dev, const AmpCompactArgs* compact,
                                  amp_stream_t stream);

/* One env-step of the hot path (motion sample + sim AMP obs / history / policy obs + dones + reset ids +
 * task reward + style reward) issued by ONE call: amp_env_step_with_reference(all phases) followed by
 * amp_disc_style_reward_prescaled_compact on bufs->disc_input / bufs->reward -- or, when bufs->disc_input is NULL (no fused
 * scaler: what a shard whose whole batch takes the fused two-layer kernel wants), by amp_disc_style_reward_compact on the rows of
 * bufs->amp_obs_buffer.  Same launches, same results; it exists
 * because four separately marshalled calls cost the host more than a small shard's step costs the GPU.  Every pointer
 * is borrowed for the duration of the enqueue; motion == NULL skips the expert-motion sample. */
typedef struct {
  const AmpEnvCfg* cfg;
  const AmpSimState* state;
  const AmpEnvBuffers* bufs;    /* needs reward; disc_input (fused scaler, amp_disc_input_layout of `disc`) or NULL (raw rows) */
  int64_t num_envs;
  const AmpMotion* motion;      /* expert-motion sample of the step, or NULL */
  const double* times;          /* dev [n_samples] */
  const int64_t* motion_ids;    /* dev [n_samples] or NULL */
  int64_t n_samples;
  int32_t K;
  int32_t reserved;
  float* expert_out;            /* dev [n_samples, K * D] */
  const AmpDisc* disc;
  float reward_scale, task_weight, style_weight;
  int32_t reserved2;
  float* logits;                /* dev [num_envs] or NULL */
  float* style;                 /* dev [num_envs] or NULL */
  float* combined;              /* dev [num_envs] or NULL */
  void* workspace;              /* amp_disc_workspace_bytes(disc, num_envs) */
  const AmpCompactArgs* compact;
} AmpHotStepArgs;
int amp_hot_step(const AmpHotStepArgs* args, amp_stream_t stream);

/* Current weights in torch.nn.Linear layout (W1 [h1, in_dim], ... , W3 [1, h2]) into caller-owned device buffers. */
int amp_disc_get_weights(const AmpDisc* h, float* w1_dev, float* b1_dev, float* w2_dev, float* b2_dev, float* w3_dev,
                         float* b3_dev, amp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Discriminator TRAINING step (skrl AMP._update, discriminator part -- third-party,
 * parity unpinned; hyper-parameters agents/skrl_g1_walk_amp_cfg.yaml:70,87-95):
 *   scaler update (train=True) per batch -> BCE(policy U replay -> 0, motion -> 1) + logit regularisation +
 *   gradient penalty w.r.t. the scaled motion states + weight decay, x loss scale -> Adam on the handle's weights.
 * ------------------------------------------------------------------------------------------------ */
typedef struct AmpDiscTrainer AmpDiscTrainer;
typedef struct {
  int64_t max_rows_per_group;   /* capacity: rows of each of the three batches (discriminator_batch_size 4096) */
  float learning_rate;          /* 5e-5 */
  float beta1, beta2, adam_epsilon; /* torch.optim.Adam defaults 0.9, 0.999, 1e-8 */
  float loss_scale;             /* discriminator_loss_scale 5.0 */
  float logit_reg_scale;        /* discriminator_logit_regularization_scale 0.05 */
  float grad_penalty_scale;     /* discriminator_gradient_penalty_scale 5.0 */
  float weight_decay_scale;     /* discriminator_weight_decay_scale 1e-4 */
  float scaler_epsilon, scaler_clip; /* RunningStandardScaler 1e-8, 5.0 */
  int32_t use_scaler;           /* 0: feed raw observations */
  int32_t update_scaler;        /* 1: train=True statistics update with every batch (fp64) */
  int32_t apply_update;         /* 0: compute loss / gradients only (Adam moments still advance with lr = 0) */
  int32_t gemm_f16x3;           /* 0 (also what a zero-initialised struct gets): every GEMM on the fp32 matrix pipe; 1: six of the backward's
                                 * ten products (dH1, gW2, gW1 of the prediction loss; a1, gW2's second product, da2 of the gradient penalty)
                                 * at fp32 accuracy on the fp16 matrix pipe (two fp16 planes per operand, three MFMAs per product, like the
                                 * inference path), planes written once per step with one power-of-two scale per operand; the forward stays
                                 * on the fp32 pipe.  Needs the forked step (gradient penalty on, hidden widths multiples of 128), else the
                                 * fp32 products run */
  int32_t defer_refresh;        /* 0 (default): every step ends by refreshing what the attached AmpDisc derives from its weights / scaler
                                 * for INFERENCE (fp16 planes, plane scales, fp32 scaler vectors: ~45 us of small launches); 1: the
                                 * steps leave them stale (the handle is then IN BETWEEN -- trained biases, old weight planes -- and
                                 * must not score) and the caller runs amp_disc_trainer_refresh() once before the next style-reward
                                 * call -- skrl's AMP._update runs 12 training steps between two rollouts */
  int32_t reserved;
} AmpDiscTrainCfg;

/* The trainer updates `disc`'s weights (and, with use_scaler, its scaler) in place; `disc` must outlive it.
 * running_mean / running_variance (device fp64 [in_dim], may be NULL = 0 / 1) and current_count seed the statistics. */
int amp_disc_trainer_create(AmpDisc* disc, const AmpDiscTrainCfg* cfg, const double* running_mean_dev,
                            const double* running_variance_dev, double current_count, amp_stream_t stream,
                            AmpDiscTrainer** out);
int amp_disc_trainer_destroy(AmpDiscTrainer* t);
/* Shape admission and tile plan of the step's weight-gradient products dW = dY^T X (the "TT" GEMM: C[M, N] (+)= A^T W, A [K, M]
 * pitch lda, W [K, N] pitch ldw, C pitch ldc; `split` != 0: split-K scratch available).  Pure host arithmetic, no GPU needed:
 * AMP_OK + the (bm x bn) tile and the k-slice count the step would launch, or AMP_ERR_INVALID when the shape is refused -- the
 * kernel has no row / column guards, so only shapes its tile covers exactly (M % bm == 0, N % bn == 0, pitches >= widths) pass.
 * (skrl's AMP._update is third-party; shapes: agents/skrl_g1_walk_amp_cfg.yaml:31-39, discriminator_batch_size :91.) */
int amp_disc_train_tt_plan(int32_t M, int32_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int32_t split, int32_t* bm,
                           int32_t* bn, int32_t* slices);
/* With cfg.defer_refresh: bring the attached AmpDisc's inference-side derived data up to date with the trained weights / scaler.
 * RULE: at most ONE refresh of a handle in flight (this call, a non-deferred amp_disc_train_step, amp_disc_set_weights: issue them
 * on one stream, or order the streams) -- the refresh kernel keeps a workgroup ticket in the handle's range record between launches;
 * amp_disc_set_weights and amp_disc_trainer_create re-zero it, so a faulted launch does not poison later refreshes. */
int amp_disc_trainer_refresh(AmpDiscTrainer* t, amp_stream_t stream);
/* Copies the running statistics (fp64 [in_dim]) into caller-owned device buffers; *count (host) = samples seen. */
int amp_disc_trainer_scaler(const AmpDiscTrainer* t, double* mean_out_dev, double* var_out_dev, double* count,
                            amp_stream_t stream);
/* Copies the optimizer state into caller-owned device buffers: first / second Adam moments of (W1, b1, W2, b2, W3, b3)
 * concatenated in their logical shapes (the layout of amp_disc_train_step's grads_dev; either may be NULL); *step (host, may be
 * NULL) = optimizer steps taken (one small blocking read-back).  What a multi-rank update compares across replicas
 * (train.py:183-196: one agent replica per GPU, kept in step) and what a checkpoint of the agent holds
 * (torch.optim.Adam.state_dict, third-party). */
int amp_disc_trainer_adam_state(const AmpDiscTrainer* t, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t* step,
                                amp_stream_t stream);
/* One step on three batches of `rows` raw AMP observations each ([rows, in_dim], row stride in elements).
 * loss_dev (may be NULL): [5] = prediction, gradient penalty, logit regularisation, weight decay (unscaled terms) and
 * [4] = loss_scale * their sum (skrl's discriminator_loss).  grads_dev (may be NULL): dL/d(W1, b1, W2, b2, W3, b3) concatenated, logical shapes.
 * Asynchronous on `stream`; part of the step (the gradient-penalty chain) is enqueued on a stream the trainer owns and joined back
 * before the step's last kernels, so the step is ordered like a single-stream enqueue for whatever follows on `stream`, and a
 * stream capture of it stays valid. */
int amp_disc_train_step(AmpDiscTrainer* t, const float* policy_dev, const float* replay_dev, const float* motion_dev,
                        int64_t rows, int64_t row_stride, float* loss_dev, float* grads_dev, amp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Device ring buffers of AMP observation rows: skrl AMP's `reply_buffer` (1 M rows) and
 * `motion_dataset` (200 k rows) -- RandomMemory objects [third-party: parity unpinned], sizes
 * agents/skrl_g1_walk_amp_cfg.yaml:44-58.  add_samples = amp_ring_append (write head wraps), sample(batch) =
 * amp_ring_sample (uniform with replacement over the rows written so far; counter-based Philox draw keyed by
 * (seed, draw, row): reproducible, no host RNG).  Appends must be serialised by the caller (one stream).
 * ------------------------------------------------------------------------------------------------ */
typedef struct AmpRing AmpRing;
int amp_ring_create(int64_t capacity_rows, int32_t row_dim, AmpRing** out);   /* capacity < 2^32 */
int amp_ring_destroy(AmpRing* r);
int64_t amp_ring_size(const AmpRing* r);   /* rows currently valid (min(total appended, capacity)) */
int64_t amp_ring_head(const AmpRing* r);   /* next write position */
int amp_ring_append(AmpRing* r, const float* rows_dev, int64_t n, int64_t row_stride, amp_stream_t stream);
/* out_dev [n, out_stride]; indices_dev [n] optional (the storage rows that were drawn).  Row i of the call takes the variate of
 * counter (first_row + i, draw): `first_row` is where this call's rows sit in the minibatch they belong to -- 0 for a whole
 * minibatch; rank w of a multi-rank discriminator update draws its n = batch / world rows with first_row = w * n, so a minibatch
 * consumes the same variates whatever the world size (the reference's --distributed mode keeps one memory per rank,
 * train.py:183-196; skrl's RandomMemory.sample is third-party: parity unpinned). */
int amp_ring_sample(const AmpRing* r, uint64_t seed, uint64_t draw, int64_t first_row, int64_t n, float* out_dev,
                    int64_t out_stride, int64_t* indices_dev, amp_stream_t stream);

/* out_dev[i] = rows_dev[pi(first + i)], i < count, where pi is a pseudo-random PERMUTATION of [0, n_rows) keyed by (seed, epoch) and
 * evaluated point-wise (6-round Feistel network + cycle walking: no sort, no index array): the epoch shuffle of an agent update's
 * rollout rows (skrl memory.sample_all + mini-batch split, agents/skrl_g1_walk_amp_cfg.yaml:65-66; third-party: parity unpinned) --
 * minibatch m of an epoch takes positions [m * per, m * per + batch): `batch` distinct rows, disjoint from the other minibatches'.
 * indices_dev [count] optional.  first + count <= n_rows. */
int amp_rows_take_permuted(const float* rows_dev, int64_t n_rows, int64_t row_stride, int32_t row_dim, uint64_t seed, uint64_t epoch,
                           int64_t first, int64_t count, float* out_dev, int64_t out_stride, int64_t* indices_dev, amp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * CSV -> npz motion converter  (replaces motions/data_convert.py:161-390: 30 -> 60 fps
 * up-sampling, forward kinematics, finite-difference + Gaussian-smoothed velocities, quaternion-difference angular
 * velocities).  The kinematic model is the URDF's joint tree (tools/urdf_to_kinematics.py), host arrays, joints in
 * parent-before-child order.
 * ------------------------------------------------------------------------------------------------ */
typedef struct AmpConverter AmpConverter;
typedef struct {
  int32_t n_joints;          /* all joints of the tree, fixed ones included (<= 64) */
  int32_t n_dof;             /* joint-angle columns of the CSV (29 for the G1) */
  int32_t n_bodies;          /* links to record */
  int32_t reserved;
  const int32_t* parent;     /* [n_joints] joint whose child link is this joint's parent link, -1 = root link */
  const int32_t* qidx;       /* [n_joints] angle column of a revolute joint, -1 = fixed */
  const double* origin_rot;  /* [n_joints, 9] row-major rotation of the joint origin (URDF rpy) */
  const double* origin_xyz;  /* [n_joints, 3] */
  const double* axis;        /* [n_joints, 3] unit axis of a revolute joint */
  const int32_t* body_joint; /* [n_bodies] joint whose child link is the body, -1 = root link */
} AmpKinModel;
typedef struct {             /* device outputs, N = 2 * n_rows - 1 frames (the npz schema, motions/README.md:11-21) */
  double* dof_positions;           /* [N, n_dof] */
  double* dof_velocities;          /* [N, n_dof] */
  float* body_positions;           /* [N, n_bodies, 3] */
  float* body_rotations;           /* [N, n_bodies, 4] (w, x, y, z) */
  float* body_linear_velocities;   /* [N, n_bodies, 3] */
  float* body_angular_velocities;  /* [N, n_bodies, 3] */
} AmpConvertOutputs;
int amp_converter_create(const AmpKinModel* model, AmpConverter** out);
int amp_converter_destroy(AmpConverter* c);
int64_t amp_convert_workspace_bytes(const AmpConverter* c, int64_t n_rows);
/* csv_dev [n_rows, 7 + n_dof] float32: root xyz, root quaternion (x, y, z, w), joint angles, sampled at 30 fps.
 * numpy1_promotion selects which numpy generation's scalar arithmetic the angular-velocity step reproduces
 * (0: numpy >= 2, the reference's G1_walk.npz; 1: numpy < 2, its custom_motion.npz). */
int amp_convert_motion(const AmpConverter* c, const float* csv_dev, int64_t n_rows, int32_t n_cols, int32_t fps_out,
                       int32_t numpy1_promotion, const AmpConvertOutputs* out, void* workspace_dev, amp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AMP_ENGINE_H */
