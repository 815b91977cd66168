// Per-step env kernel: done mask (+ per-tile reset counts), task reward, AMP feature extraction with the
// in-place K-frame history shift, policy observation (+ optional actor history with reset warm-start).
// One pass, any subset of the three phases.  gfx950 only; -ffp-contract=off.
//
// Work decomposition: a workgroup (256 lanes) owns a TILE of 64 consecutive envs.
//   stage   the tile's [64, n_dof] rows of joint_pos / joint_vel (/ actions / joint_acc) are contiguous in
//           HBM: lanes walk them flat (coalesced) into the LDS observation tile.
//   per-env wave 0, one env per lane: root-body features (quat_apply x2), done bits, wave ballot ->
//           reset count of the tile.  The four reward reductions run one (env, term) per lane on all 4 waves.
//   write   every output (AMP buffer rows, policy obs, actor history) is walked flat from the LDS tile, so
//           stores are contiguous runs of D (or P) floats per env.
#include "amp_common.hpp"
#include "motion_kernels.hpp"

namespace amp {


struct EnvPlan {
  int32_t n_dof, dof_pad, n_key, D, Db, K;
  int32_t n_actor, use_last_actions, use_command, hist_actions, hist_command;
  int32_t P, Pcur, per;
  int32_t early_termination, reward_mode;
  uint32_t phases;
  int64_t max_episode_length;
  float termination_height;
  float s_term, s_act, s_lim, s_acc, s_vel;
  float w_track, sigma_sq, thr, val_at_thr, slope;
};

// `block` = index of the env tile, `smem` = the workgroup's dynamic LDS: a device function so that it can run either
// as its own kernel or as one half of the horizontally fused launch below.
template <int kTileEnvs>
__device__ __forceinline__ void env_step_body(const EnvPlan& p, const AmpSimState& st, const AmpEnvBuffers& bf, int64_t N,
                                              int64_t block, float* smem) {
  const int D = p.D, nd = p.n_dof, ndp = p.dof_pad;
  float* s_obs = smem;                          // [64, D]
  float* s_act = s_obs + kTileEnvs * D;         // [64, ndp]   (reward)
  float* s_acc = s_act + kTileEnvs * ndp;       // [64, ndp]   (reward)
  float* s_red = s_acc + kTileEnvs * ndp;       // [4, 64]     (reward)
  int* s_flag = reinterpret_cast<int*>(s_red + 4 * kTileEnvs);  // [64] just_reset
  float* s_lim = reinterpret_cast<float*>(s_flag + kTileEnvs);  // [64 | 1, 2*nd + 1] soft joint limits (reward)
  const int lim_row = 2 * nd + 1;                                // odd stride: conflict-free one-env-per-lane reads

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile_base = block * kTileEnvs;
  const int n_tile = (int)((N - tile_base) < kTileEnvs ? (N - tile_base) : kTileEnvs);
  const bool do_dones = p.phases & AMP_PHASE_DONES;
  const bool do_rew = p.phases & AMP_PHASE_REWARD;
  const bool do_obs = p.phases & AMP_PHASE_OBS;
  const bool g1_rew = do_rew && p.reward_mode == 1;

  // ---- stage the contiguous per-DoF rows of the tile into LDS -----------------------------------
  if (do_obs || g1_rew) {
    for (int e = tid; e < n_tile * nd; e += kBlock) {
      const int s = e / nd, j = e - s * nd;
      const int64_t env = tile_base + s;
      s_obs[s * D + j] = st.joint_pos[env * st.joint_pos_stride + j];
      s_obs[s * D + nd + j] = st.joint_vel[env * st.joint_vel_stride + j];
    }
  }
  if (g1_rew) {
    for (int e = tid; e < n_tile * nd; e += kBlock) {
      const int s = e / nd, j = e - s * nd;
      const int64_t env = tile_base + s;
      s_act[s * ndp + j] = st.actions[env * st.actions_stride + j];
      s_acc[s * ndp + j] = st.joint_acc[env * st.joint_acc_stride + j];
    }
    // soft limits: one shared [nd, 2] row (stride 0) or a [64, nd, 2] block, staged with coalesced loads so the
    // reduction below never waits on a global load per DoF
    const int lim_rows = st.soft_limits_stride == 0 ? 1 : n_tile;
    for (int e = tid; e < lim_rows * 2 * nd; e += kBlock) {
      const int s = e / (2 * nd), c = e - s * 2 * nd;
      s_lim[s * lim_row + c] = st.soft_limits[(tile_base + s) * st.soft_limits_stride + c];
    }
  }

  // ---- per-env work, wave 0, one env per lane -----------------------------------------------------
  int died = 0;
  if (wave == 0) {
    int reset_bit = 0;
    if (lane < n_tile) {
      const int64_t env = tile_base + lane;
      if (do_dones) {
        // g1_amp_env.py:321-330
        const int tout = st.episode_length[env] >= p.max_episode_length - 1;
        died = p.early_termination ? (st.root_pos[env * st.root_pos_stride + 2] < p.termination_height) : 0;
        bf.died[env] = (uint8_t)died;
        bf.time_out[env] = (uint8_t)tout;
        reset_bit = died | tout;
        if (bf.reset_mask) bf.reset_mask[env] = (uint8_t)reset_bit;
      } else if (g1_rew) {
        died = bf.died[env];
      }
      if (do_obs) {
        // compute_obs features that are not plain copies (g1_amp_env.py:545-555)
        const float* rp = st.root_pos + env * st.root_pos_stride;
        const float* rq = st.root_quat + env * st.root_quat_stride;
        const float* rl = st.root_lin_vel + env * st.root_lin_vel_stride;
        const float* ra = st.root_ang_vel + env * st.root_ang_vel_stride;
        const float px = rp[0], py = rp[1], pz = rp[2];
        const Quat q{rq[0], rq[1], rq[2], rq[3]};
        const Vec3 tg = quat_apply_ref(q, Vec3{1.0f, 0.0f, 0.0f});
        const Vec3 nm = quat_apply_ref(q, Vec3{0.0f, 0.0f, 1.0f});
        float* o = s_obs + lane * D + 2 * nd;
        o[0] = pz;
        o[1] = tg.x; o[2] = tg.y; o[3] = tg.z;
        o[4] = nm.x; o[5] = nm.y; o[6] = nm.z;
        o[7] = rl[0]; o[8] = rl[1]; o[9] = rl[2];
        o[10] = ra[0]; o[11] = ra[1]; o[12] = ra[2];
        const float* bp = st.body_pos + env * st.body_pos_stride;
        for (int k = 0; k < p.n_key; ++k) {
          const float* kp = bp + (int64_t)st.key_body[k] * 3;
          o[13 + 3 * k + 0] = kp[0] - px;
          o[13 + 3 * k + 1] = kp[1] - py;
          o[13 + 3 * k + 2] = kp[2] - pz;
        }
        if (p.n_actor > 1) s_flag[lane] = bf.just_reset[env];
      }
    }
    if (do_dones && bf.reset_tile_counts) {
      const unsigned long long b = __ballot(reset_bit);
      if (lane == 0) bf.reset_tile_counts[block] = __popcll(b);
    }
  }
  __syncthreads();

  // ---- task reward -------------------------------------------------------------------------------
  if (do_rew) {
    if (p.reward_mode == 0) {
      if (wave == 0 && lane < n_tile) bf.reward[tile_base + lane] = 1.0f;  // humanoid_amp_env.py:128-129
    } else {
      // compute_rewards (g1_amp_env.py:564-606): wave w reduces term w of env `lane` over the DoFs
      float acc = 0.0f;
      if (lane < n_tile) {
        if (wave == 0) {
          for (int j = 0; j < nd; ++j) { const float a = s_act[lane * ndp + j]; acc += a * a; }
        } else if (wave == 1) {
          const float* lim = s_lim + (st.soft_limits_stride == 0 ? 0 : lane * lim_row);
          for (int j = 0; j < nd; ++j) {
            const float x = s_obs[lane * D + j];
            float o = -fminf(x - lim[2 * j], 0.0f);
            o += fmaxf(x - lim[2 * j + 1], 0.0f);
            acc += o;
          }
        } else if (wave == 2) {
          for (int j = 0; j < nd; ++j) { const float a = s_acc[lane * ndp + j]; acc += a * a; }
        } else {
          for (int j = 0; j < nd; ++j) { const float a = s_obs[lane * D + nd + j]; acc += a * a; }
        }
      }
      if (lane < kTileEnvs) s_red[wave * kTileEnvs + lane] = acc;
      __syncthreads();
      if (wave == 0 && lane < n_tile) {
        const int64_t env = tile_base + lane;
        const float r_term = p.s_term * (float)died;
        const float r_act = p.s_act * s_red[lane];
        const float r_lim = p.s_lim * s_red[kTileEnvs + lane];
        const float r_acc = p.s_acc * s_red[2 * kTileEnvs + lane];
        const float r_vel = p.s_vel * s_red[3 * kTileEnvs + lane];
        const float basic = (((r_term + r_act) + r_lim) + r_acc) + r_vel;
        float track = 0.0f, err = 0.0f;
        if (p.use_command) {
          // g1_amp_env.py:249-265: planar body-frame velocity error, exp reward with linear floor (:500-532)
          const float* rq = st.root_quat + env * st.root_quat_stride;
          const float* rl = st.root_lin_vel + env * st.root_lin_vel_stride;
          const Vec3 vb = quat_rotate_inverse_ref(Quat{rq[0], rq[1], rq[2], rq[3]}, Vec3{rl[0], rl[1], rl[2]});
          const float dx = vb.x - st.command[env * 2 + 0];
          const float dy = vb.y - st.command[env * 2 + 1];
          err = sqrtf(dx * dx + dy * dy);
          const float e2 = err * err;
          const float lin = p.val_at_thr - p.slope * (e2 - p.thr);
          const float ex = p.w_track * expf(-e2 / p.sigma_sq);
          track = e2 > p.thr ? lin : ex;
        }
        const float total = basic + track;
        bf.reward[env] = total;
        if (bf.reward_terms) {
          float* t = bf.reward_terms + env;
          t[0 * N] = total; t[1 * N] = track; t[2 * N] = err; t[3 * N] = r_term;
          t[4 * N] = r_act; t[5 * N] = r_lim; t[6 * N] = r_acc; t[7 * N] = r_vel;
        }
      }
    }
  }

  // ---- observations ------------------------------------------------------------------------------
  if (do_obs) {
    // AMP history, in place: slot k+1 <- slot k (k = K-2..0), slot 0 <- obs (g1_amp_env.py:187-190).
    // A lane owns column j of env s for every slot, so no other lane touches what it reads or writes.
    const int K = p.K;
    const int64_t rowK = (int64_t)K * D;
    float* buf = bf.amp_obs_buffer + tile_base * rowK;
    const int count = n_tile * D;
    const float inv_d = 1.0f / (float)D;
    constexpr int U = 4;  // independent columns per lane per trip: their loads are issued before any store
    // optional fused discriminator input: the same values, scaled, into disc_input (fp32 rows or fp16 (p0, p1) pairs:
    // one 32-bit store per element either way)
    const bool pairs = bf.disc_input_format == AMP_DISC_INPUT_F16_PAIRS;
    const bool fused = bf.disc_input != nullptr;
    uint32_t* const xs = reinterpret_cast<uint32_t*>(bf.disc_input) + tile_base * bf.disc_input_stride;
    const float s_x = bf.disc_plane_scale;
    const float* const mu = bf.scaler_mean;
    const float* const dn = bf.scaler_den;
    const float clip = bf.scaler_clip;
    auto emit = [&](int64_t off, float v, int c) {  // same operations, in the same order, as disc.hip's scaler passes
      if (mu) {
        v = (v - mu[c]) / dn[c];  // skrl RunningStandardScaler, exact fp32 divide
        v = fminf(fmaxf(v, -clip), clip);
      }
      xs[off] = pairs ? plane_pair(v * s_x) : __float_as_uint(v);
    };
    for (int e0 = tid; e0 < count; e0 += U * kBlock) {
      float* col[U];
      int64_t xoff[U];
      int jc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int e = e0 + u * kBlock;
        e = e < count ? e : count - 1;
        const int s = (int)(((float)e + 0.5f) * inv_d);  // exact: e < 64 * D
        jc[u] = e - s * D;
        col[u] = buf + s * rowK + jc[u];
        xoff[u] = s * bf.disc_input_stride + jc[u];
      }
      for (int hi = K - 2; hi >= 0; hi -= 2) {
        float h[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (hi - i >= 0) h[u][i] = col[u][(int64_t)(hi - i) * D];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (hi - i >= 0 && e0 + u * kBlock < count) {
              col[u][(int64_t)(hi - i + 1) * D] = h[u][i];
              if (fused) emit(xoff[u] + (hi - i + 1) * D, h[u][i], (hi - i + 1) * D + jc[u]);
            }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (e0 + u * kBlock < count) {
          const float v = s_obs[e0 + u * kBlock];
          col[u][0] = v;
          if (fused) emit(xoff[u], v, jc[u]);
        }
    }
    // policy observation (g1_amp_env.py:195-242; humanoid_amp_env.py:126)
    const int P = p.P, Pcur = p.Pcur, Db = p.Db;
    float* pol = bf.policy_obs + tile_base * P;
    for (int e = tid; e < n_tile * Pcur; e += kBlock) {
      const int s = e / Pcur, c = e - s * Pcur;
      const int64_t env = tile_base + s;
      float v;
      if (!p.use_last_actions) v = s_obs[s * D + c];
      else if (c < Db) v = s_obs[s * D + c];
      else if (c < Db + nd) v = st.last_actions[env * nd + (c - Db)];
      else v = st.command[env * 2 + (c - Db - nd)];
      pol[(int64_t)s * P + c] = v;
    }
    if (p.n_actor > 1) {
      const int per = p.per, H = p.n_actor - 1;
      for (int e = tid; e < n_tile * per; e += kBlock) {
        const int s = e / per, c = e - s * per;
        const int64_t env = tile_base + s;
        float hv;
        if (c < Db) hv = s_obs[s * D + c];
        else if (p.hist_actions && c < Db + nd) hv = st.last_actions[env * nd + (c - Db)];
        else hv = st.command[env * 2 + (c - Db - (p.hist_actions ? nd : 0))];
        float* hb = bf.actor_history + (env * H) * per + c;
        float* po = pol + (int64_t)s * P + Pcur + c;
        if (s_flag[s]) {
          // warm start: a freshly reset env fills every history slot with its first frame (:216-222)
          for (int i = 0; i < H; ++i) { hb[(int64_t)i * per] = hv; po[(int64_t)i * per] = hv; }
        } else {
          for (int i = H - 2; i >= 0; --i) {
            const float x = hb[(int64_t)i * per];
            hb[(int64_t)(i + 1) * per] = x;
            po[(int64_t)(i + 1) * per] = x;
          }
          hb[0] = hv;
          po[0] = hv;
        }
      }
      if (wave == 0 && lane < n_tile) bf.just_reset[tile_base + lane] = 0;
    }
  }
}

template <int kTileEnvs>
__global__ __launch_bounds__(kBlock) void env_step_kernel(EnvPlan p, AmpSimState st, AmpEnvBuffers bf, int64_t N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  env_step_body<kTileEnvs>(p, st, bf, N, (int64_t)blockIdx.x, smem);
}

// Horizontally fused launch: workgroups [0, env_blocks) run the env step, the rest the expert-motion sample
// (collect_reference_motions: no data dependence on the env state), so the two byte-moving kernels of a step share one
// launch and overlap instead of running back to back.  Bit-identical to the two separate launches.
struct ExpertArgs {
  MotionView v;
  const double* times;
  const int64_t* ids;
  int64_t n;
  int32_t K;
  float* out;
};
template <int kTileEnvs>
__global__ __launch_bounds__(kBlock) void env_step_reference_kernel(EnvPlan p, AmpSimState st, AmpEnvBuffers bf, int64_t N,
                                                                    unsigned env_blocks, ExpertArgs x) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.x < env_blocks) env_step_body<kTileEnvs>(p, st, bf, N, (int64_t)blockIdx.x, smem);
  else collect_reference_body(x.v, x.times, x.ids, x.n, x.K, x.out, nullptr, nullptr, (int64_t)(blockIdx.x - env_blocks), smem);
}

static int make_plan(const AmpEnvCfg* c, uint32_t phases, EnvPlan* p) {
  AMP_REQUIRE(c->n_dof >= 1 && c->n_dof <= 256, "amp_env_step: n_dof %d out of range", c->n_dof);
  AMP_REQUIRE(c->n_key >= 1 && c->n_key <= kMaxKey, "amp_env_step: n_key %d out of range", c->n_key);
  AMP_REQUIRE(c->num_amp_observations >= 1, "amp_env_step: num_amp_observations must be >= 1");
  AMP_REQUIRE(c->num_actor_observations >= 1, "amp_env_step: num_actor_observations must be >= 1");
  AMP_REQUIRE(c->reward_mode == 0 || c->reward_mode == 1, "amp_env_step: unknown reward_mode %d", c->reward_mode);
  p->n_dof = c->n_dof;
  p->dof_pad = c->n_dof | 1;  // odd LDS row stride: conflict-free one-env-per-lane reads
  p->n_key = c->n_key;
  p->D = 2 * c->n_dof + 13 + 3 * c->n_key;
  p->Db = p->D - 3 * c->n_key;
  p->K = c->num_amp_observations;
  p->n_actor = c->num_actor_observations;
  p->use_last_actions = c->use_last_actions != 0;
  p->use_command = c->use_command != 0;
  p->hist_actions = c->history_include_last_actions != 0;
  p->hist_command = c->history_include_command != 0;
  if (!p->use_last_actions) {
    AMP_REQUIRE(p->n_actor == 1, "amp_env_step: actor history needs use_last_actions");
    p->Pcur = p->D;  // humanoid: the policy sees the full AMP frame
    p->per = 0;
  } else {
    p->Pcur = p->Db + c->n_dof + (p->use_command ? 2 : 0);
    p->per = p->n_actor > 1 ? p->Db + (p->hist_actions ? c->n_dof : 0) + ((p->hist_command && p->use_command) ? 2 : 0) : 0;
  }
  p->P = p->Pcur + (p->n_actor - 1) * p->per;
  p->early_termination = c->early_termination != 0;
  p->reward_mode = c->reward_mode;
  p->phases = phases;
  p->max_episode_length = c->max_episode_length;
  p->termination_height = c->termination_height;
  p->s_term = c->rew_termination;
  p->s_act = c->rew_action_l2;
  p->s_lim = c->rew_joint_pos_limits;
  p->s_acc = c->rew_joint_acc_l2;
  p->s_vel = c->rew_joint_vel_l2;
  // exp_reward_with_floor scalars are python floats (fp64) in the reference, rounded to fp32 only when
  // they meet a tensor (g1_amp_env.py:516-529)
  const double sigma_sq = c->track_sigma * c->track_sigma;
  const double thr = c->track_floor * sigma_sq;
  p->w_track = (float)c->rew_track_vel;
  p->sigma_sq = (float)sigma_sq;
  p->thr = (float)thr;
  p->val_at_thr = (float)(c->rew_track_vel * exp(-c->track_floor));
  p->slope = (float)(c->rew_track_vel / sigma_sq * exp(-c->track_floor));
  return AMP_OK;
}

}  // namespace amp

using namespace amp;

extern "C" {

int32_t amp_env_step_tile_envs(int64_t num_envs) {
  // measured (MI355X, G1 K=2): 65 536 envs 72 / 78 / 91 us with 64 / 32 / 16-env tiles; 16 384 envs 39 / 27 / 23 us
  // envs per workgroup: 64 when that still gives >= 1024 workgroups (4 per CU), else smaller tiles so that a small
  // shard is spread over the whole chip (the kernel is latency-bound there: one 64-env tile takes ~25 us alone)
  if (num_envs >= 64 * 1024) return 64;
  if (num_envs >= 32 * 1024) return 32;
  return 16;
}

int64_t amp_policy_obs_size(const AmpEnvCfg* cfg) {
  EnvPlan p;
  if (!cfg || make_plan(cfg, 0, &p) != AMP_OK) return -1;
  return p.P;
}

int64_t amp_actor_history_frame_size(const AmpEnvCfg* cfg) {
  EnvPlan p;
  if (!cfg || make_plan(cfg, 0, &p) != AMP_OK) return -1;
  return p.per;
}

static int env_step_launch(const AmpEnvCfg* cfg, const AmpSimState* st, const AmpEnvBuffers* bf, int64_t N, uint32_t phases,
                           const ExpertArgs* expert, amp_stream_t stream) {
  AMP_REQUIRE(cfg && st && bf, "amp_env_step: null argument");
  AMP_REQUIRE(N >= 0, "amp_env_step: negative num_envs");
  AMP_REQUIRE(phases != 0 && (phases & ~7u) == 0, "amp_env_step: phases must be a non-empty OR of AMP_PHASE_*");
  if (N == 0) return AMP_OK;
  EnvPlan p;
  int rc = make_plan(cfg, phases, &p);
  if (rc != AMP_OK) return rc;
  const bool g1_rew = (phases & AMP_PHASE_REWARD) && p.reward_mode == 1;
  if (phases & AMP_PHASE_DONES) {
    AMP_REQUIRE(st->episode_length && bf->died && bf->time_out, "amp_env_step(dones): null buffer");
    AMP_REQUIRE(!p.early_termination || st->root_pos, "amp_env_step(dones): root_pos is null");
  }
  if (phases & AMP_PHASE_REWARD) {
    AMP_REQUIRE(bf->reward, "amp_env_step(reward): reward buffer is null");
    if (g1_rew) {
      AMP_REQUIRE(st->joint_pos && st->joint_vel && st->joint_acc && st->actions && st->soft_limits && bf->died,
                  "amp_env_step(reward): null buffer");
      AMP_REQUIRE(!p.use_command || (st->command && st->root_quat && st->root_lin_vel),
                  "amp_env_step(reward): velocity tracking needs command, root_quat and root_lin_vel");
    }
  }
  if (phases & AMP_PHASE_OBS) {
    AMP_REQUIRE(st->joint_pos && st->joint_vel && st->root_pos && st->root_quat && st->root_lin_vel && st->root_ang_vel &&
                    st->body_pos && bf->amp_obs_buffer && bf->policy_obs,
                "amp_env_step(obs): null buffer");
    AMP_REQUIRE(!p.use_last_actions || st->last_actions, "amp_env_step(obs): last_actions is null");
    AMP_REQUIRE(!(p.use_last_actions && p.use_command) || st->command, "amp_env_step(obs): command is null");
    AMP_REQUIRE(p.n_actor == 1 || (bf->actor_history && bf->just_reset), "amp_env_step(obs): actor history buffers are null");
    for (int k = 0; k < p.n_key; ++k) AMP_REQUIRE(st->key_body[k] >= 0, "amp_env_step(obs): negative key body index");
    AMP_REQUIRE(!bf->disc_input || bf->disc_input_stride >= (int64_t)p.K * p.D, "amp_env_step(obs): disc_input_stride too small");
    AMP_REQUIRE(!bf->disc_input || !bf->scaler_mean || bf->scaler_den, "amp_env_step(obs): scaler_den is null");
    AMP_REQUIRE(!bf->disc_input || bf->disc_input_format == AMP_DISC_INPUT_F32_ROWS ||
                    (bf->disc_input_format == AMP_DISC_INPUT_F16_PAIRS && bf->disc_plane_scale > 0.0f),
                "amp_env_step(obs): bad disc_input format / plane scale");
  }
  const bool per_env_limits = g1_rew && st->soft_limits_stride != 0;
  const int tile = amp_env_step_tile_envs(N);
  const size_t lds = sizeof(float) * ((size_t)tile * p.D + 2 * (size_t)tile * p.dof_pad + 4 * tile) + sizeof(int) * tile +
                     sizeof(float) * (size_t)(per_env_limits ? tile : 1) * (2 * p.n_dof + 1);
  AMP_REQUIRE(lds <= 64 * 1024, "amp_env_step: observation tile needs %zu B of LDS (> 64 KiB)", lds);
  const unsigned grid = (unsigned)((N + tile - 1) / tile);
  if (expert && expert->n > 0) {
    const size_t lds_x = expert_lds(expert->v.D);
    const size_t lds_f = lds > lds_x ? lds : lds_x;
    AMP_REQUIRE(lds_f <= 64 * 1024, "amp_env_step_with_reference: tile needs %zu B of LDS (> 64 KiB)", lds_f);
    const unsigned grid_x = (unsigned)((expert->n * expert->K + kExpertTile - 1) / kExpertTile);
    amp::TraceScope trace__("env_step_reference_kernel", (hipStream_t)stream);
    if (tile == 64) env_step_reference_kernel<64><<<grid + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, grid, *expert);
    else if (tile == 32) env_step_reference_kernel<32><<<grid + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, grid, *expert);
    else env_step_reference_kernel<16><<<grid + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, grid, *expert);
    return launch_status("env_step_reference_kernel");
  }
  { amp::TraceScope trace__("env_step_kernel", (hipStream_t)stream);
    if (tile == 64) env_step_kernel<64><<<grid, kBlock, lds, (hipStream_t)stream>>>(p, *st, *bf, N);
    else if (tile == 32) env_step_kernel<32><<<grid, kBlock, lds, (hipStream_t)stream>>>(p, *st, *bf, N);
    else env_step_kernel<16><<<grid, kBlock, lds, (hipStream_t)stream>>>(p, *st, *bf, N);
  }
  return launch_status("env_step_kernel");
}

int amp_env_step(const AmpEnvCfg* cfg, const AmpSimState* st, const AmpEnvBuffers* bf, int64_t N, uint32_t phases,
                 amp_stream_t stream) {
  return env_step_launch(cfg, st, bf, N, phases, nullptr, stream);
}

int amp_env_step_with_reference(const AmpEnvCfg* cfg, const AmpSimState* st, const AmpEnvBuffers* bf, int64_t N, uint32_t phases,
                                const AmpMotion* motion, const double* times_dev, const int64_t* motion_ids_dev, int64_t n_samples,
                                int32_t K, float* expert_out_dev, amp_stream_t stream) {
  AMP_REQUIRE(motion, "amp_env_step_with_reference: null motion handle");
  AMP_REQUIRE(motion->has_layout, "amp_env_step_with_reference: call amp_motion_set_obs_layout first");
  AMP_REQUIRE(n_samples >= 0 && K >= 1, "amp_env_step_with_reference: bad sample count / K");
  AMP_REQUIRE(n_samples == 0 || (times_dev && expert_out_dev), "amp_env_step_with_reference: null buffer");
  ExpertArgs x{motion->v, times_dev, motion_ids_dev, n_samples, K, expert_out_dev};
  return env_step_launch(cfg, st, bf, N, phases, &x, stream);
}

}  // extern "C"
