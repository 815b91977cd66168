// Per-step env kernel: done mask (+ per-tile reset counts), task reward, AMP feature extraction with the
// in-place K-frame history shift, policy observation (+ optional actor history with reset warm-start).
// One pass, any subset of the three phases.  gfx950 only; -ffp-contract=off.
//
// Work decomposition: a workgroup (256 lanes) owns a TILE of 32 / 16 consecutive envs (amp_env_step_tile_envs).
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

// Row index of flat element e of a [rows, width] block (e * width^-1 is far from an integer boundary: e < 2^16).
__device__ __forceinline__ int row_of(int e, float inv_width) { return (int)(((float)e + 0.5f) * inv_width); }

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
  // scaler statistics of the fused discriminator input, staged once: the write loops then read LDS instead of waiting on
  // two global loads per element
  float* s_mu = s_lim + ((p.phases & AMP_PHASE_REWARD) && p.reward_mode == 1 && st.soft_limits_stride != 0 ? kTileEnvs : 1) * lim_row;
  float* s_dn = s_mu + p.K * D;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile_base = block * kTileEnvs;
  const int n_tile = (int)((N - tile_base) < kTileEnvs ? (N - tile_base) : kTileEnvs);
  const bool do_dones = p.phases & AMP_PHASE_DONES;
  const bool do_rew = p.phases & AMP_PHASE_REWARD;
  const bool do_obs = p.phases & AMP_PHASE_OBS;
  const bool g1_rew = do_rew && p.reward_mode == 1;

  if (do_obs && bf.disc_input != nullptr && bf.scaler_mean != nullptr)
    for (int c = tid; c < p.K * D; c += kBlock) { s_mu[c] = bf.scaler_mean[c]; s_dn[c] = bf.scaler_den[c]; }
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
    // optional fused discriminator input: the same values, scaled, into disc_input (fp32 rows: one 32-bit store per
    // element; fp16 plane blocks: per row and 32-column k-block [p0 x 32 | p1 x 32] halves, two 16-bit stores here)
    const bool blocks = bf.disc_input_format == AMP_DISC_INPUT_F16_BLOCKS;
    const bool fused = bf.disc_input != nullptr;
    uint32_t* const xs = reinterpret_cast<uint32_t*>(bf.disc_input) + tile_base * bf.disc_input_stride;
    const float s_x = bf.disc_plane_scale;
    const bool scaled = bf.scaler_mean != nullptr;
    const float clip = bf.scaler_clip;
    auto emit = [&](int64_t off, float v, int c) {  // same operations, in the same order, as disc.hip's scaler passes
      if (scaled) {
        v = (v - s_mu[c]) / s_dn[c];  // skrl RunningStandardScaler, exact fp32 divide
        v = fminf(fmaxf(v, -clip), clip);
      }
      if (blocks) {
        const uint32_t w = plane_pair(v * s_x);  // p0 | p1 << 16
        uint16_t* h = reinterpret_cast<uint16_t*>(xs) + 2 * (off - c) + (c >> 5) * 64 + (c & 31);  // row start + block + column
        h[0] = (uint16_t)w;
        h[32] = (uint16_t)(w >> 16);
      } else {
        xs[off] = __float_as_uint(v);
      }
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

// ------------------------------------------------------------------------------------------------
// Fast tile body -- the hot-path configuration: all three phases, K == 2, no actor history, a whole tile, per-DoF
// inputs whose [T, n_dof] blocks are contiguous and 16-B aligned (all checked on the host; anything else runs the
// generic body above).  Same arithmetic, bit for bit; what differs is how the bytes move:
//   * every HBM read of the tile is issued before the first wait (16-B flat loads of the per-DoF blocks, wave 0's
//     per-env values, the old history slot of the columns a lane owns, the scaler statistics);
//   * the tile's new AMP rows are assembled as an LDS image [T, 2*D] -- exactly the tile's contiguous span of the AMP
//     buffer -- and streamed out with 16-B stores; the discriminator input and the policy observation are written
//     two columns per lane (8-B stores, one row/column split per pair).
// The generic body spends ~40 VALU instructions per output float on index arithmetic and 4-B stores and is
// VALU-bound (no stores at all: 36 us of 58 at 65 536 envs); this one is bounded by the memory system.
// ------------------------------------------------------------------------------------------------
constexpr int kFastVec = 2;  // 16-B loads per lane per [T, n_dof] block: T * n_dof <= 2 * 4 * kBlock

template <int T>
__device__ __forceinline__ void env_step_fast_body(const EnvPlan& p, const AmpSimState& st, const AmpEnvBuffers& bf,
                                                   int64_t N, int64_t block, float* smem) {
  constexpr int kHist = (T * 96) / kBlock;  // history columns per lane (D <= 96)
  const int D = p.D, nd = p.n_dof, ndp = p.dof_pad, KD = 2 * p.D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile_base = block * T;
  const bool g1 = p.reward_mode == 1;
  const bool extra = p.use_last_actions;  // policy obs = [obs[:Db] | last_actions | command]
  const bool has_cmd = extra && p.use_command;
  const bool per_env_limits = g1 && st.soft_limits_stride != 0;
  const bool fused = bf.disc_input != nullptr;
  const bool scaled = fused && bf.scaler_mean != nullptr;
  const int lim_row = 2 * nd + 1;

  float* s_img = smem;                    // [T, 2*D]  the tile's new AMP rows: slot 0 = this step, slot 1 = old slot 0
  float* s_act = s_img + T * KD;          // [T, ndp]  (reward)
  float* s_acc = s_act + T * ndp;         // [T, ndp]  (reward)
  float* s_la = s_acc + T * ndp;          // [T, nd]   last_actions rows (policy obs)
  float* s_cmd = s_la + T * nd;           // [T, 2]    command rows
  float* s_red = s_cmd + T * 2;           // [4, T]    reward partial sums
  float* s_mu = s_red + 4 * T;            // [2*D]     scaler mean
  float* s_dn = s_mu + KD;                // [2*D]     scaler sqrt(var) + eps
  float* s_lim = s_dn + KD;               // [T | 1, 2*nd + 1]

  // ---- issue every HBM read of the tile ---------------------------------------------------------------
  const int nvec = T * nd / 4;
  f4 v_pos[kFastVec], v_vel[kFastVec], v_act[kFastVec], v_acc[kFastVec], v_last[kFastVec], v_cmd;
  {
    const f4* g_pos = reinterpret_cast<const f4*>(st.joint_pos + tile_base * nd);
    const f4* g_vel = reinterpret_cast<const f4*>(st.joint_vel + tile_base * nd);
    const f4* g_act = reinterpret_cast<const f4*>(st.actions + tile_base * nd);
    const f4* g_acc = reinterpret_cast<const f4*>(st.joint_acc + tile_base * nd);
    const f4* g_last = reinterpret_cast<const f4*>(st.last_actions + tile_base * nd);
#pragma unroll
    for (int v = 0; v < kFastVec; ++v) {
      const int i = tid + v * kBlock;
      if (i < nvec) {
        v_pos[v] = g_pos[i];
        v_vel[v] = g_vel[i];
        if (g1) { v_act[v] = g_act[i]; v_acc[v] = g_acc[i]; }
        if (extra) v_last[v] = g_last[i];
      }
    }
    if (has_cmd && tid < T / 2) v_cmd = reinterpret_cast<const f4*>(st.command + tile_base * 2)[tid];
  }
  const bool env_lane = wave == 0 && lane < T;
  const int64_t env = tile_base + lane;
  int64_t ep_len;
  float rp[3], rq[4], rl[3], ra[3], kb[kMaxKey][3], cmd[2];
  if (env_lane) {
    ep_len = st.episode_length[env];
    const float* g = st.root_pos + env * st.root_pos_stride;
    rp[0] = g[0]; rp[1] = g[1]; rp[2] = g[2];
    const float* gq = st.root_quat + env * st.root_quat_stride;
    rq[0] = gq[0]; rq[1] = gq[1]; rq[2] = gq[2]; rq[3] = gq[3];
    const float* gl = st.root_lin_vel + env * st.root_lin_vel_stride;
    rl[0] = gl[0]; rl[1] = gl[1]; rl[2] = gl[2];
    const float* ga = st.root_ang_vel + env * st.root_ang_vel_stride;
    ra[0] = ga[0]; ra[1] = ga[1]; ra[2] = ga[2];
    const float* bp = st.body_pos + env * st.body_pos_stride;
#pragma unroll
    for (int k = 0; k < kMaxKey; ++k)
      if (k < p.n_key) {
        const float* kp = bp + (int64_t)st.key_body[k] * 3;
        kb[k][0] = kp[0]; kb[k][1] = kp[1]; kb[k][2] = kp[2];
      }
    if (g1 && p.use_command) { cmd[0] = st.command[env * 2 + 0]; cmd[1] = st.command[env * 2 + 1]; }
  }
  float* const buf = bf.amp_obs_buffer + tile_base * KD;
  const float inv_d = 1.0f / (float)D;
  float h[kHist];
#pragma unroll
  for (int u = 0; u < kHist; ++u) {
    const int e = tid + u * kBlock;
    if (e < T * D) {
      const int s = row_of(e, inv_d);
      h[u] = buf[s * KD + (e - s * D)];
    }
  }
  float mu_c, dn_c;
  if (scaled && tid < KD) { mu_c = bf.scaler_mean[tid]; dn_c = bf.scaler_den[tid]; }  // KD <= 192 < kBlock

  // ---- registers -> LDS ----------------------------------------------------------------------------------
  const float inv_nd = 1.0f / (float)nd;
#pragma unroll
  for (int v = 0; v < kFastVec; ++v) {
    const int i = tid + v * kBlock;
    if (i < nvec) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int e = 4 * i + c;
        const int s = row_of(e, inv_nd), j = e - s * nd;
        s_img[s * KD + j] = v_pos[v][c];
        s_img[s * KD + nd + j] = v_vel[v][c];
        if (g1) { s_act[s * ndp + j] = v_act[v][c]; s_acc[s * ndp + j] = v_acc[v][c]; }
      }
      if (extra) reinterpret_cast<f4*>(s_la)[i] = v_last[v];
    }
  }
  if (has_cmd && tid < T / 2) reinterpret_cast<f4*>(s_cmd)[tid] = v_cmd;
  if (g1) {
    const int lim_rows = per_env_limits ? T : 1;
    for (int e = tid; e < lim_rows * 2 * nd; e += kBlock) {
      const int s = e / (2 * nd), c = e - s * 2 * nd;
      s_lim[s * lim_row + c] = st.soft_limits[(tile_base + s) * st.soft_limits_stride + c];
    }
  }
  int died = 0;
  if (wave == 0) {
    int reset_bit = 0;
    if (env_lane) {
      // g1_amp_env.py:321-330
      const int tout = ep_len >= p.max_episode_length - 1;
      died = p.early_termination ? (rp[2] < p.termination_height) : 0;
      bf.died[env] = (uint8_t)died;
      bf.time_out[env] = (uint8_t)tout;
      reset_bit = died | tout;
      if (bf.reset_mask) bf.reset_mask[env] = (uint8_t)reset_bit;
      // compute_obs features that are not plain copies (g1_amp_env.py:545-555)
      const Quat q{rq[0], rq[1], rq[2], rq[3]};
      const Vec3 tg = quat_apply_ref(q, Vec3{1.0f, 0.0f, 0.0f});
      const Vec3 nm = quat_apply_ref(q, Vec3{0.0f, 0.0f, 1.0f});
      float* o = s_img + lane * KD + 2 * nd;
      o[0] = rp[2];
      o[1] = tg.x; o[2] = tg.y; o[3] = tg.z;
      o[4] = nm.x; o[5] = nm.y; o[6] = nm.z;
      o[7] = rl[0]; o[8] = rl[1]; o[9] = rl[2];
      o[10] = ra[0]; o[11] = ra[1]; o[12] = ra[2];
#pragma unroll
      for (int k = 0; k < kMaxKey; ++k)
        if (k < p.n_key) {
          o[13 + 3 * k + 0] = kb[k][0] - rp[0];
          o[13 + 3 * k + 1] = kb[k][1] - rp[1];
          o[13 + 3 * k + 2] = kb[k][2] - rp[2];
        }
    }
    if (bf.reset_tile_counts) {
      const unsigned long long b = __ballot(reset_bit);
      if (lane == 0) bf.reset_tile_counts[block] = __popcll(b);
    }
  }
#pragma unroll
  for (int u = 0; u < kHist; ++u) {
    const int e = tid + u * kBlock;
    if (e < T * D) {
      const int s = row_of(e, inv_d);
      s_img[s * KD + D + (e - s * D)] = h[u];  // slot 1 <- old slot 0 (g1_amp_env.py:187-190)
    }
  }
  if (scaled && tid < KD) { s_mu[tid] = mu_c; s_dn[tid] = dn_c; }
  __syncthreads();

  // ---- task reward -------------------------------------------------------------------------------
  if (!g1) {
    if (env_lane) bf.reward[env] = 1.0f;  // humanoid_amp_env.py:128-129
  } else {
    // compute_rewards (g1_amp_env.py:564-606): wave w reduces term w of env `lane` over the DoFs
    float acc = 0.0f;
    if (lane < T) {
      if (wave == 0) {
        for (int j = 0; j < nd; ++j) { const float a = s_act[lane * ndp + j]; acc += a * a; }
      } else if (wave == 1) {
        const float* lim = s_lim + (per_env_limits ? lane * lim_row : 0);
        for (int j = 0; j < nd; ++j) {
          const float x = s_img[lane * KD + j];
          float o = -fminf(x - lim[2 * j], 0.0f);
          o += fmaxf(x - lim[2 * j + 1], 0.0f);
          acc += o;
        }
      } else if (wave == 2) {
        for (int j = 0; j < nd; ++j) { const float a = s_acc[lane * ndp + j]; acc += a * a; }
      } else {
        for (int j = 0; j < nd; ++j) { const float a = s_img[lane * KD + nd + j]; acc += a * a; }
      }
      s_red[wave * T + lane] = acc;
    }
    __syncthreads();
    if (env_lane) {
      const float r_term = p.s_term * (float)died;
      const float r_act = p.s_act * s_red[lane];
      const float r_lim = p.s_lim * s_red[T + lane];
      const float r_acc = p.s_acc * s_red[2 * T + lane];
      const float r_vel = p.s_vel * s_red[3 * T + lane];
      const float basic = (((r_term + r_act) + r_lim) + r_acc) + r_vel;
      float track = 0.0f, err = 0.0f;
      if (p.use_command) {
        // g1_amp_env.py:249-265: planar body-frame velocity error, exp reward with linear floor (:500-532)
        const Vec3 vb = quat_rotate_inverse_ref(Quat{rq[0], rq[1], rq[2], rq[3]}, Vec3{rl[0], rl[1], rl[2]});
        const float dx = vb.x - cmd[0];
        const float dy = vb.y - cmd[1];
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

  // ---- outputs: LDS image -> HBM ----------------------------------------------------------------------------
  {  // AMP buffer: the tile's rows are one contiguous 16-B aligned span
    const f4* img4 = reinterpret_cast<const f4*>(s_img);
    f4* dst4 = reinterpret_cast<f4*>(buf);
    for (int i = tid; i < T * KD / 4; i += kBlock) dst4[i] = img4[i];
  }
  if (fused) {
    // the same rows, scaled, as the discriminator's input: two columns per lane (rows hold an even number of them)
    const bool blocks = bf.disc_input_format == AMP_DISC_INPUT_F16_BLOCKS;
    uint32_t* const xs = reinterpret_cast<uint32_t*>(bf.disc_input) + tile_base * bf.disc_input_stride;
    const float s_x = bf.disc_plane_scale, clip = bf.scaler_clip;
    const float inv_ch = 1.0f / (float)D;  // D two-column items per row
#pragma unroll 4
    for (int it = tid; it < T * D; it += kBlock) {
      const int s = row_of(it, inv_ch), c = 2 * (it - s * D);
      const float2 v = *reinterpret_cast<const float2*>(s_img + s * KD + c);
      float x0 = v.x, x1 = v.y;
      if (scaled) {  // same operations, in the same order, as disc.hip's scaler passes
        const float2 m = *reinterpret_cast<const float2*>(s_mu + c);
        const float2 d = *reinterpret_cast<const float2*>(s_dn + c);
        x0 = (x0 - m.x) / d.x;  // skrl RunningStandardScaler, exact fp32 divide
        x1 = (x1 - m.y) / d.y;
        x0 = fminf(fmaxf(x0, -clip), clip);
        x1 = fminf(fmaxf(x1, -clip), clip);
      }
      if (blocks) {
        // columns c, c + 1 (c even: the same k-block): their p0 halves are one 32-bit word of the block's first 64 B,
        // their p1 halves the word 64 B further
        const uint32_t wa = plane_pair(x0 * s_x), wb = plane_pair(x1 * s_x);  // p0 | p1 << 16 of each column
        uint32_t* blk = xs + s * bf.disc_input_stride + (c >> 5) * 32 + ((c & 31) >> 1);
        blk[0] = __builtin_amdgcn_perm(wb, wa, 0x05040100u);   // [p0(c), p0(c + 1)]
        blk[16] = __builtin_amdgcn_perm(wb, wa, 0x07060302u);  // [p1(c), p1(c + 1)]
      } else {
        uint2 o;
        o.x = __float_as_uint(x0);
        o.y = __float_as_uint(x1);
        *reinterpret_cast<uint2*>(xs + s * bf.disc_input_stride + c) = o;
      }
    }
  }
  {  // policy observation (g1_amp_env.py:195-242; humanoid_amp_env.py:126), P == Pcur (no actor history here)
    const int P = p.P, Db = p.Db;
    float* pol = bf.policy_obs + tile_base * P;
    auto value = [&](int s, int c) -> float {
      if (!extra || c < Db) return s_img[s * KD + c];
      if (c < Db + nd) return s_la[s * nd + (c - Db)];
      return s_cmd[s * 2 + (c - Db - nd)];
    };
    if ((P & 1) == 0) {
      const int half = P >> 1;
      const float inv_half = 1.0f / (float)half;
#pragma unroll 4
      for (int it = tid; it < T * half; it += kBlock) {
        const int s = row_of(it, inv_half), c = 2 * (it - s * half);
        float2 o;
        o.x = value(s, c);
        o.y = value(s, c + 1);
        *reinterpret_cast<float2*>(pol + (int64_t)s * P + c) = o;
      }
    } else {
      const float inv_p = 1.0f / (float)P;
      for (int e = tid; e < T * P; e += kBlock) {
        const int s = row_of(e, inv_p);
        pol[e] = value(s, e - s * P);
      }
    }
  }
}

template <int kTileEnvs>
__global__ __launch_bounds__(kBlock) void env_step_kernel(EnvPlan p, AmpSimState st, AmpEnvBuffers bf, int64_t N,
                                                          unsigned block0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  env_step_body<kTileEnvs>(p, st, bf, N, (int64_t)(block0 + blockIdx.x), smem);
}

template <int T>
__global__ __launch_bounds__(kBlock) void env_step_fast_kernel(EnvPlan p, AmpSimState st, AmpEnvBuffers bf, int64_t N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  env_step_fast_body<T>(p, st, bf, N, (int64_t)blockIdx.x, smem);
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
template <int T>
__global__ __launch_bounds__(kBlock) void env_step_fast_reference_kernel(EnvPlan p, AmpSimState st, AmpEnvBuffers bf, int64_t N,
                                                                         unsigned env_blocks, ExpertArgs x) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.x < env_blocks) env_step_fast_body<T>(p, st, bf, N, (int64_t)blockIdx.x, smem);
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
  // envs per workgroup.  Generic body, measured (MI355X, G1 K=2): 65 536 envs 72 / 78 / 91 us with 64 / 32 / 16-env
  // tiles; 16 384 envs 39 / 27 / 23 us.  The fast body (32-env tile: 34 KB of LDS, 4 workgroups per CU) takes 42 us at
  // 65 536 envs, so large shards use 32; small shards are spread over the whole chip with 16 (latency-bound there).
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
                    (bf->disc_input_format == AMP_DISC_INPUT_F16_BLOCKS && bf->disc_plane_scale > 0.0f && bf->disc_input_stride % 32 == 0),
                "amp_env_step(obs): bad disc_input format / plane scale");
  }
  const bool per_env_limits = g1_rew && st->soft_limits_stride != 0;
  const int tile = amp_env_step_tile_envs(N);
  const size_t lds = sizeof(float) * ((size_t)tile * p.D + 2 * (size_t)tile * p.dof_pad + 4 * tile) + sizeof(int) * tile +
                     sizeof(float) * (size_t)(per_env_limits ? tile : 1) * (2 * p.n_dof + 1) +
                     sizeof(float) * 2 * (size_t)p.K * p.D;  // + the scaler statistics of the fused discriminator input
  AMP_REQUIRE(lds <= 64 * 1024, "amp_env_step: observation tile needs %zu B of LDS (> 64 KiB)", lds);
  const unsigned grid = (unsigned)((N + tile - 1) / tile);
  const bool with_expert = expert && expert->n > 0;
  const size_t lds_x = with_expert ? expert_lds(expert->v.D) : 0;
  const unsigned grid_x = with_expert ? (unsigned)((expert->n * expert->K + kExpertTile - 1) / kExpertTile) : 0;
  AMP_REQUIRE(lds_x <= 64 * 1024, "amp_env_step_with_reference: tile needs %zu B of LDS (> 64 KiB)", lds_x);
  const char* label = with_expert ? "env_step_reference_kernel" : "env_step_kernel";

  // The hot-path configuration runs the fast tile body on every whole tile (see env_step_fast_body); a ragged last
  // tile, and every other configuration, runs the generic body.
  auto aligned = [](const void* ptr, uintptr_t a) { return (reinterpret_cast<uintptr_t>(ptr) & (a - 1)) == 0; };
  auto rows16 = [&](const float* ptr, int64_t stride) { return stride == p.n_dof && aligned(ptr, 16); };
  const size_t lds_fast = sizeof(float) * ((size_t)tile * 2 * p.D + 2 * (size_t)tile * p.dof_pad + (size_t)tile * p.n_dof + 6 * (size_t)tile +
                                           4 * (size_t)p.D + (size_t)(per_env_limits ? tile : 1) * (2 * p.n_dof + 1));
  bool fast = phases == (AMP_PHASE_DONES | AMP_PHASE_REWARD | AMP_PHASE_OBS) && p.K == 2 && p.n_actor == 1 && p.D <= 96 && tile <= 32 && N >= tile &&
              (int64_t)tile * p.n_dof <= (int64_t)kFastVec * 4 * kBlock && lds_fast <= 64 * 1024;
  fast = fast && rows16(st->joint_pos, st->joint_pos_stride) && rows16(st->joint_vel, st->joint_vel_stride);
  if (fast && p.reward_mode == 1) fast = rows16(st->actions, st->actions_stride) && rows16(st->joint_acc, st->joint_acc_stride);
  if (fast && p.use_last_actions) fast = aligned(st->last_actions, 16) && (!p.use_command || aligned(st->command, 16));
  fast = fast && aligned(bf->amp_obs_buffer, 16) && aligned(bf->policy_obs, 8);
  if (fast && bf->disc_input) fast = aligned(bf->disc_input, 8) && (bf->disc_input_stride & 1) == 0;
  if (fast) {
    const unsigned full = (unsigned)(N / tile);
    const size_t lds_f = lds_fast > lds_x ? lds_fast : lds_x;
    {
      amp::TraceScope trace__(label, (hipStream_t)stream);
      if (with_expert) {
        if (tile == 32) env_step_fast_reference_kernel<32><<<full + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, full, *expert);
        else env_step_fast_reference_kernel<16><<<full + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, full, *expert);
      } else {
        if (tile == 32) env_step_fast_kernel<32><<<full, kBlock, lds_fast, (hipStream_t)stream>>>(p, *st, *bf, N);
        else env_step_fast_kernel<16><<<full, kBlock, lds_fast, (hipStream_t)stream>>>(p, *st, *bf, N);
      }
    }
    if (full != grid) {  // ragged last tile
      if (tile == 32) env_step_kernel<32><<<1, kBlock, lds, (hipStream_t)stream>>>(p, *st, *bf, N, full);
      else env_step_kernel<16><<<1, kBlock, lds, (hipStream_t)stream>>>(p, *st, *bf, N, full);
    }
    return launch_status(label);
  }
  if (with_expert) {
    const size_t lds_f = lds > lds_x ? lds : lds_x;
    amp::TraceScope trace__(label, (hipStream_t)stream);
    if (tile == 32) env_step_reference_kernel<32><<<grid + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, grid, *expert);
    else env_step_reference_kernel<16><<<grid + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, grid, *expert);
    return launch_status(label);
  }
  { amp::TraceScope trace__(label, (hipStream_t)stream);
    if (tile == 32) env_step_kernel<32><<<grid, kBlock, lds, (hipStream_t)stream>>>(p, *st, *bf, N, 0u);
    else env_step_kernel<16><<<grid, kBlock, lds, (hipStream_t)stream>>>(p, *st, *bf, N, 0u);
  }
  return launch_status(label);
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
