// Per-step env kernel: done mask (+ per-tile reset counts), task reward, AMP feature extraction with the
// in-place K-frame history shift, policy observation (+ optional actor history with reset warm-start).
// One pass, any subset of the three phases.  gfx950 only; -ffp-contract=off.
//
// Work decomposition: a workgroup (256 lanes) owns a TILE of 32 / 16 / 8 consecutive envs (amp_env_step_tile_envs).
// Two bodies, bit-identical where both apply:
//   * the GENERIC body (any phase subset, actor history, strided inputs, ragged last tile): lanes walk the tile's
//     [tile, n_dof] rows flat into an LDS observation tile; wave 0, one env per lane, does the root-body features, done
//     bits and the wave ballot -> reset count; the four reward reductions run one (env, term) per lane on all 4 waves;
//     every output is walked flat from the LDS tile (contiguous runs of D or P floats per env);
//   * the DMA tile body (the hot-path configuration: all three phases, no actor history, whole tiles): inputs by LDS-DMA,
//     outputs walked column-major -- see env_step_dma_pass.
// With an expert-sample request the launch also carries the expert tiles (collect_reference_*_body, motion_kernels.hpp).
#include "amp_common.hpp"
#include "motion_kernels.hpp"
#include <type_traits>

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
// DMA tile body -- any phase subset (the hot path launches all three at once; the DirectRLEnv hooks launch DONES | REWARD
// before the reset and OBS after it), any K >= 1, with or without actor history, whole tiles (checked on the host; anything else runs the
// generic body).  Same arithmetic as the generic body, bit for bit.  The generic body
// spends ~40 VALU instructions per output float on row/column splits, 64-bit addressing and 4-B stores and is bound by
// instruction issue, and so was its first replacement (registers -> LDS scatter, ~1 900 VALU instructions per lane
// around 86 KB of traffic per tile: PMC showed the SIMD issue slots ~94 % busy; profiles/r02_k_pmc_per_kernel.md).
// This one is built to issue few instructions per byte:
//   * inputs go HBM -> LDS by LDS-DMA (global_load_lds): per env row the old history slots 0..K-2 land one slot further
//     down the row's LDS image, joint_pos / joint_vel at its head (dword pieces), actions / joint_acc / last_actions /
//     command / scaler statistics as flat 16-B pieces.  No VGPR staging, no ds_write, no per-element row/column split;
//   * outputs are walked COLUMN-major: a lane owns one column pair (its scaler statistics, block offset and source
//     array are per-lane constants) and steps down the tile's rows by adding a pitch;
//   * the fp16 planes of a column pair come from v_cvt_pk_f16_f32 (two columns per convert, already in store order).
// A workgroup owns T = amp_env_step_tile_envs(cfg, N) envs: 32 / 16 as above, 8 where the [T, K*D] image would not fit
// 64 KB of LDS (K = 10).  Measured (65 536 envs, launch alone / behind a cache flush): K = 2 42.7 / 63.2 -> 36.8 / 57.1 us,
// K = 10 273 -> 149 us, humanoid 35.2 / 51.4 -> 30.7 / 48.0 us.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* env_gptr_t;
typedef __attribute__((address_space(3))) void* env_lptr_t;
typedef _Float16 env_h2 __attribute__((ext_vector_type(2)));
typedef float env_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void dma4(const float* g, float* l) {   // 64 lanes x 4 B -> LDS l + lane * 4 (l wave-uniform)
  __builtin_amdgcn_global_load_lds((env_gptr_t)g, (env_lptr_t)l, 4, 0, 0);
}
__device__ __forceinline__ void dma16(const float* g, float* l) {  // 64 lanes x 16 B -> LDS l + lane * 16
  __builtin_amdgcn_global_load_lds((env_gptr_t)g, (env_lptr_t)l, 16, 0, 0);
}
// the same with the non-temporal hint (gfx940+ cache policy bit 1): per-step state that no later kernel reads again
__device__ __forceinline__ void dma4_nt(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((env_gptr_t)g, (env_lptr_t)l, 4, 0, 2);
}
__device__ __forceinline__ void dma16_nt(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((env_gptr_t)g, (env_lptr_t)l, 16, 0, 2);
}

// floats of LDS one tile of T envs needs (host and device agree through this one function)
// row pitch of the tile's LDS image: the whole [K*D] AMP row when observations are written, else just joint_pos | joint_vel
// (+ 1: odd pitch, one-env-per-lane reads of the reward reductions stay conflict-free)
__host__ __device__ inline int env_dma_row_pitch(int KD, int nd, bool obs) { return obs ? KD : ((2 * nd + 4) & ~3) + 1; }
__host__ __device__ inline int env_dma_lds_floats(int T, int KD, int nd, bool per_env_limits, bool obs = true) {
  const int ndT = (T * nd + 3) & ~3;
  return ((T * env_dma_row_pitch(KD, nd, obs) + 3) & ~3) + 3 * ndT + 2 * T + 4 * T + 2 * ((KD + 3) & ~3) + (per_env_limits ? T : 1) * (2 * nd + 1);
}

#ifdef AMP_ENV_TIMELINE  // diagnostic builds only (tools/env_timeline.py): per-workgroup phase stamps, 100 MHz wall clock
__device__ unsigned long long* g_env_timeline;  // [g_env_timeline_rows][8]
__device__ unsigned g_env_timeline_rows;        // capacity in workgroups: workgroups past it are not stamped
#define AMP_ENV_STAMP(slot)                                                                                              \
  do {                                                                                                                   \
    if (g_env_timeline && threadIdx.x == 0 && blockIdx.x < g_env_timeline_rows)                                         \
      g_env_timeline[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();                              \
  } while (0)
extern "C" int amp_debug_env_timeline(unsigned long long* buf, unsigned rows) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_env_timeline_rows), &rows, sizeof(rows)) != hipSuccess) return -2;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_env_timeline), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#else
#define AMP_ENV_STAMP(slot) do {} while (0)
#endif

template <int T>
__device__ __forceinline__ void env_step_dma_pass(const EnvPlan& p, const AmpSimState& st, const AmpEnvBuffers& bf,
                                                  int64_t N, int64_t block, float* smem) {
  const int64_t tile_base = block * T;
  const int D = p.D, nd = p.n_dof, KD = p.K * p.D, C = KD - D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the wave that does the per-env work rotates with the tile index: consecutive workgroups of a CU put it on different SIMDs
  const int role = (wave + (int)block) & 3;
  // phase subset of this launch (wave-uniform): the hot path asks for all three, the DirectRLEnv hooks for DONES | REWARD
  // before the reset and OBS after it
  const bool do_dones = p.phases & AMP_PHASE_DONES, do_rew = p.phases & AMP_PHASE_REWARD, do_obs = p.phases & AMP_PHASE_OBS;
  const bool g1 = do_rew && p.reward_mode == 1;
  const bool extra = do_obs && p.use_last_actions;  // policy obs = [obs[:Db] | last_actions | command]
  const bool has_cmd = extra && p.use_command;
  const bool per_env_limits = g1 && st.soft_limits_stride != 0;
  const bool fused = do_obs && bf.disc_input != nullptr;
  const bool scaled = fused && bf.scaler_mean != nullptr;
  const int lim_row = 2 * nd + 1;
  const int ndT = (T * nd + 3) & ~3;
  const int RP = env_dma_row_pitch(KD, nd, do_obs);  // == KD whenever observations are written

  AMP_ENV_STAMP(0);
  float* s_img = smem;                     // [T, K*D]  the tile's new AMP rows (== its span of the AMP buffer)
  float* s_act = s_img + ((T * RP + 3) & ~3);  // [T, nd]   flat copies (reward / policy obs)
  float* s_acc = s_act + ndT;
  float* s_la = s_acc + ndT;
  float* s_cmd = s_la + ndT;               // [T, 2]
  float* s_red = s_cmd + 2 * T;            // [4, T]    reward partial sums
  float* s_mu = s_red + 4 * T;             // [K*D]     scaler mean
  float* s_dn = s_mu + ((KD + 3) & ~3);    // [K*D]     scaler sqrt(var) + eps
  float* s_lim = s_dn + ((KD + 3) & ~3);   // [T | 1, 2*nd + 1]

  // ---- HBM -> LDS, no registers in between -----------------------------------------------------------------
  // (staging the 4-B granular pieces -- history and joint rows -- through registers instead, one column per lane, was
  //  measured equal at K = 2 and 13 % slower at K = 10)
  float* const buf = bf.amp_obs_buffer + tile_base * KD;
#ifndef AMP_ENV_XP_NO_ROW_DMA   // (diagnostic builds only: upper bound of what fewer row pieces could buy; results are wrong)
  if (do_obs || g1) {
#else
  if (false) {
#endif
#pragma unroll 1
  for (int r = wave; r < T; r += 4) {
    float* row = s_img + r * RP;
    if (do_obs) {
#pragma unroll 1
      for (int c0 = 0; c0 < C; c0 += 64)  // slot k + 1 <- old slot k (g1_amp_env.py:187-190)
        if (c0 + lane < C) dma4_nt(buf + r * KD + c0 + lane, row + D + c0);
    }
    const float* gp = st.joint_pos + (tile_base + r) * st.joint_pos_stride;
    const float* gv = st.joint_vel + (tile_base + r) * st.joint_vel_stride;
#pragma unroll 1
    for (int c0 = 0; c0 < nd; c0 += 64)
      if (c0 + lane < nd) {
        dma4_nt(gp + c0 + lane, row + c0);
        dma4_nt(gv + c0 + lane, row + nd + c0);
      }
  }
  }
  {
    const int n16 = T * nd / 4;  // T * nd is a multiple of 4 (T >= 8)
#pragma unroll 1
    for (int pc = wave * 64; pc < n16; pc += kBlock) {
      const int i = pc + lane;
      if (i < n16) {
        if (g1) {
          dma16_nt(st.actions + tile_base * nd + 4 * i, s_act + 4 * pc);
          dma16_nt(st.joint_acc + tile_base * nd + 4 * i, s_acc + 4 * pc);
        }
        if (extra) dma16_nt(st.last_actions + tile_base * nd + 4 * i, s_la + 4 * pc);
      }
    }
    if (has_cmd && wave == 3 && lane < T / 2) dma16(st.command + tile_base * 2 + 4 * lane, s_cmd);
    if (scaled) {
#pragma unroll 1
      for (int pc = wave * 64; pc < KD; pc += kBlock)
        if (pc + lane < KD) {
          dma4(bf.scaler_mean + pc + lane, s_mu + pc);
          dma4(bf.scaler_den + pc + lane, s_dn + pc);
        }
    }
  }
  if (g1) {
    if (per_env_limits) {
#pragma unroll 1
      for (int r = wave; r < T; r += 4) {
        const float* gl = st.soft_limits + (tile_base + r) * st.soft_limits_stride;
#pragma unroll 1
        for (int c0 = 0; c0 < 2 * nd; c0 += 64)
          if (c0 + lane < 2 * nd) dma4(gl + c0 + lane, s_lim + r * lim_row + c0);
      }
    } else {
      for (int e = tid; e < 2 * nd; e += kBlock) s_lim[e] = st.soft_limits[e];
    }
  }
  AMP_ENV_STAMP(1);  // every DMA piece issued
  // ---- per-env work: one env per lane of the role-0 wave, under the DMA's flight time -------------------------
  const bool env_lane = role == 0 && lane < T;
  const int64_t env = tile_base + lane;
  const bool track = g1 && p.use_command;
  int died = 0;
  float rq[4], rl[3], cmd[2];
  if (role == 0) {
    int reset_bit = 0;
    if (env_lane) {
      // every load first (an un-needed one is skipped by a wave-uniform branch), then the arithmetic
      int64_t ep_len = 0;
      float px = 0.0f, py = 0.0f, pz = 0.0f, ax = 0.0f, ay = 0.0f, az = 0.0f;
      float kb[kMaxKey][3];
      if (do_dones) ep_len = st.episode_length[env];
      if (do_obs || (do_dones && p.early_termination)) {
        const float* g = st.root_pos + env * st.root_pos_stride;
        px = g[0]; py = g[1]; pz = g[2];
      }
      if (do_obs || track) {
        const float* gq = st.root_quat + env * st.root_quat_stride;
        rq[0] = gq[0]; rq[1] = gq[1]; rq[2] = gq[2]; rq[3] = gq[3];
        const float* gl = st.root_lin_vel + env * st.root_lin_vel_stride;
        rl[0] = gl[0]; rl[1] = gl[1]; rl[2] = gl[2];
      }
      if (do_obs) {
        const float* ga = st.root_ang_vel + env * st.root_ang_vel_stride;
        ax = ga[0]; ay = ga[1]; az = ga[2];
        const float* bp = st.body_pos + env * st.body_pos_stride;
#pragma unroll
        for (int k = 0; k < kMaxKey; ++k) {  // branch-free: slots past n_key re-read key body 0 (a cache hit) and are dropped
          const float* kp = bp + (int64_t)(k < p.n_key ? st.key_body[k] : st.key_body[0]) * 3;
          kb[k][0] = kp[0]; kb[k][1] = kp[1]; kb[k][2] = kp[2];
        }
      }
      if (track) { cmd[0] = st.command[env * 2 + 0]; cmd[1] = st.command[env * 2 + 1]; }
      if (do_dones) {
        // g1_amp_env.py:321-330
        const int tout = ep_len >= p.max_episode_length - 1;
        died = p.early_termination ? (pz < p.termination_height) : 0;
        bf.died[env] = (uint8_t)died;
        bf.time_out[env] = (uint8_t)tout;
        reset_bit = died | tout;
        if (bf.reset_mask) bf.reset_mask[env] = (uint8_t)reset_bit;
      } else if (g1) {
        died = bf.died[env];  // a REWARD launch after a separate DONES launch
      }
      if (do_obs) {
        // compute_obs features that are not plain copies (g1_amp_env.py:545-555)
        const Quat q{rq[0], rq[1], rq[2], rq[3]};
        const Vec3 tg = quat_apply_ref(q, Vec3{1.0f, 0.0f, 0.0f});
        const Vec3 nm = quat_apply_ref(q, Vec3{0.0f, 0.0f, 1.0f});
        float* o = s_img + lane * KD + 2 * nd;
        o[0] = pz;
        o[1] = tg.x; o[2] = tg.y; o[3] = tg.z;
        o[4] = nm.x; o[5] = nm.y; o[6] = nm.z;
        o[7] = rl[0]; o[8] = rl[1]; o[9] = rl[2];
        o[10] = ax; o[11] = ay; o[12] = az;
#pragma unroll
        for (int k = 0; k < kMaxKey; ++k)
          if (k < p.n_key) {
            o[13 + 3 * k + 0] = kb[k][0] - px;
            o[13 + 3 * k + 1] = kb[k][1] - py;
            o[13 + 3 * k + 2] = kb[k][2] - pz;
          }
      }
    }
    if (do_dones && bf.reset_tile_counts) {
      const unsigned long long bits = __ballot(reset_bit);
      if (lane == 0) bf.reset_tile_counts[block] = __popcll(bits);
    }
  }
  AMP_ENV_STAMP(2);  // (thread 0's wave) per-env work done
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces have landed
  __syncthreads();
  AMP_ENV_STAMP(3);  // inputs landed, barrier passed

  // ---- task reward -------------------------------------------------------------------------------
  if (!g1) {
    if (do_rew && env_lane) bf.reward[env] = 1.0f;  // humanoid_amp_env.py:128-129
  } else {
    // compute_rewards (g1_amp_env.py:564-606): the wave with role w reduces term w of env `lane` over the DoFs
    float acc = 0.0f;
    if (lane < T) {
      if (role == 0) {
        for (int j = 0; j < nd; ++j) { const float a = s_act[lane * nd + j]; acc += a * a; }
      } else if (role == 1) {
        const float* lim = s_lim + (per_env_limits ? lane * lim_row : 0);
        for (int j = 0; j < nd; ++j) {
          const float x = s_img[lane * RP + j];
          float o = -fminf(x - lim[2 * j], 0.0f);
          o += fmaxf(x - lim[2 * j + 1], 0.0f);
          acc += o;
        }
      } else if (role == 2) {
        for (int j = 0; j < nd; ++j) { const float a = s_acc[lane * nd + j]; acc += a * a; }
      } else {
        for (int j = 0; j < nd; ++j) { const float a = s_img[lane * RP + nd + j]; acc += a * a; }
      }
      s_red[role * T + lane] = acc;
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

  AMP_ENV_STAMP(4);  // reward done
  // ---- outputs: LDS image -> HBM ----------------------------------------------------------------------------
  if (!do_obs) return;
  {  // AMP buffer: the tile's rows are one contiguous 16-B aligned span.  Non-temporal stores (here and for the policy
     // observation): nothing in the step reads these rows back, and the lines they would claim in the Infinity Cache hold the
     // hidden layer the GEMMs are about to stream through (measured: 328 -> 320 us per step at 65 536 envs)
    const f4* img4 = reinterpret_cast<const f4*>(s_img);
    f4* dst4 = reinterpret_cast<f4*>(buf);
    if (bf.amp_obs_read_next) {
      for (int i = tid; i < T * KD / 4; i += kBlock) dst4[i] = img4[i];
    } else {
      for (int i = tid; i < T * KD / 4; i += kBlock) __builtin_nontemporal_store(img4[i], dst4 + i);
    }
  }
  AMP_ENV_STAMP(5);  // AMP buffer stores issued
#ifdef AMP_ENV_XP_NO_DISC_WALK   // ABLATION builds only (results wrong): what the launch gives back if the discriminator's input is not produced here
  if (false) {
#else
  if (fused) {
#endif
    // the same rows, scaled, as the discriminator's input.  A lane owns the column pair (c, c + 1) -- c even, so both
    // sit in one 32-column k-block -- and walks rows r0, r0 + step, ...; `scaled` / `blocks` are compile-time inside
    // the row loop (one instantiation per combination, picked once)
    const bool blocks = bf.disc_input_format == AMP_DISC_INPUT_F16_BLOCKS;
    uint32_t* const xs = reinterpret_cast<uint32_t*>(bf.disc_input) + tile_base * bf.disc_input_stride;
    const float s_x = bf.disc_plane_scale, clip = bf.scaler_clip;
    const int64_t pitch = bf.disc_input_stride;
    auto column_pair = [&](auto scaled_c, auto blocks_c, const int c, const int r0, const int step) {
      constexpr bool kScaled = decltype(scaled_c)::value, kBlocks = decltype(blocks_c)::value;
      env_f2 m = {0.0f, 0.0f}, d = {1.0f, 1.0f};
      if (kScaled) {
        m = *reinterpret_cast<const env_f2*>(s_mu + c);
        d = *reinterpret_cast<const env_f2*>(s_dn + c);
      }
      const float* src = s_img + r0 * KD + c;
      uint32_t* dst = xs + r0 * pitch + (kBlocks ? (c >> 5) * 32 + ((c & 31) >> 1) : c);
#pragma unroll 2
      for (int r = r0; r < T; r += step, src += step * KD, dst += step * pitch) {
        const env_f2 v = *reinterpret_cast<const env_f2*>(src);
        float x0 = v.x, x1 = v.y;
        if (kScaled) {  // same operations, in the same order, as disc.hip's scaler passes
          x0 = (x0 - m.x) / d.x;  // skrl RunningStandardScaler, exact fp32 divide
          x1 = (x1 - m.y) / d.y;
          x0 = fminf(fmaxf(x0, -clip), clip);
          x1 = fminf(fmaxf(x1, -clip), clip);
        }
        if (kBlocks) {
          // p0 = rn16(v), p1 = rn16(v - p0) (plane_pair) for both columns at once: the p0 halves are one 32-bit word of
          // the block's first 64 B, the p1 halves the word 64 B further
          const env_f2 y = {x0 * s_x, x1 * s_x};
          const env_h2 a = __builtin_convertvector(y, env_h2);
          const env_f2 af = __builtin_convertvector(a, env_f2);
          const env_f2 rem = {y.x - af.x, y.y - af.y};
          const env_h2 lo = __builtin_convertvector(rem, env_h2);
          dst[0] = __builtin_bit_cast(uint32_t, a);
          dst[16] = __builtin_bit_cast(uint32_t, lo);
        } else {
          uint2 o;
          o.x = __float_as_uint(x0);
          o.y = __float_as_uint(x1);
          *reinterpret_cast<uint2*>(dst) = o;
        }
      }
    };
    const int PR = KD >> 1;  // column pairs per row (K*D is even: checked on the host)
    auto walk = [&](auto scaled_c, auto blocks_c) {
      if (PR >= kBlock) {
        for (int c2 = tid; c2 < PR; c2 += kBlock) column_pair(scaled_c, blocks_c, 2 * c2, 0, 1);
      } else {
        const int G = kBlock / PR, g = row_of(tid, 1.0f / (float)PR);
        if (g < G) column_pair(scaled_c, blocks_c, 2 * (tid - g * PR), g, G);
      }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    if (scaled) { if (blocks) walk(yes{}, yes{}); else walk(yes{}, no{}); }
    else { if (blocks) walk(no{}, yes{}); else walk(no{}, no{}); }
  }
  AMP_ENV_STAMP(6);  // discriminator input issued
  {  // policy observation (g1_amp_env.py:195-242; humanoid_amp_env.py:126): Pcur current columns of a P-float row
    const int P = p.P, Pcur = p.Pcur, Db = p.Db;
    float* pol = bf.policy_obs + tile_base * P;
    const int la_off = (int)(s_la - smem), cmd_off = (int)(s_cmd - smem);
    auto source = [&](const int c, int& src_pitch) -> const float* {  // LDS column c of the policy row (selects, no table)
      const bool in_img = !extra || c < Db, in_la = c < Db + nd;
      src_pitch = in_img ? KD : (in_la ? nd : 2);
      return smem + (in_img ? c : (in_la ? la_off + (c - Db) : cmd_off + (c - Db - nd)));
    };
    if (((P | Pcur) & 1) == 0) {
      auto column_pair = [&](const int c, const int r0, const int step) {
        int pa, pb;
        const float* a = source(c, pa);
        const float* b = source(c + 1, pb);
        a += r0 * pa;
        b += r0 * pb;
        float* dst = pol + (int64_t)r0 * P + c;
#pragma unroll 2
        for (int r = r0; r < T; r += step, a += step * pa, b += step * pb, dst += step * P) {
          env_f2 o;
          o.x = *a;
          o.y = *b;
          __builtin_nontemporal_store(o, reinterpret_cast<env_f2*>(dst));
        }
      };
      const int PR = Pcur >> 1;
      if (PR >= kBlock) {
        for (int c2 = tid; c2 < PR; c2 += kBlock) column_pair(2 * c2, 0, 1);
      } else {
        const int G = kBlock / PR, g = row_of(tid, 1.0f / (float)PR);
        if (g < G) column_pair(2 * (tid - g * PR), g, G);
      }
    } else {
      auto column = [&](const int c, const int r0, const int step) {
        int pa;
        const float* a = source(c, pa) + r0 * pa;
        float* dst = pol + (int64_t)r0 * P + c;
        for (int r = r0; r < T; r += step, a += step * pa, dst += step * P) __builtin_nontemporal_store(*a, dst);
      };
      if (Pcur >= kBlock) {
        for (int c = tid; c < Pcur; c += kBlock) column(c, 0, 1);
      } else {
        const int G = kBlock / Pcur, g = row_of(tid, 1.0f / (float)Pcur);
        if (g < G) column(tid - g * Pcur, g, G);
      }
    }
    AMP_ENV_STAMP(7);  // policy observation issued
    if (p.n_actor > 1) {
      // actor history (g1_amp_env.py:207-242): H = n_actor - 1 older frames of `per` floats behind the current columns.  A
      // lane owns one column of the frame [obs[:Db] | last_actions? | command?] (per-lane constant LDS source) and steps down
      // the tile's rows: shift the env's slots one back in place (a freshly reset env fills them all with its first frame,
      // :216-222) and mirror them into the policy row.  Same copies as the generic body.
      const int per = p.per, H = p.n_actor - 1;
      const int la_h = p.hist_actions ? nd : 0;
      auto column = [&](const int c, const int r0, const int step) {
        const bool in_img = c < Db, in_la = c < Db + la_h;
        const int pa = in_img ? KD : (in_la ? nd : 2);
        const float* a = smem + (in_img ? c : (in_la ? la_off + (c - Db) : cmd_off + (c - Db - la_h))) + r0 * pa;
        for (int r = r0; r < T; r += step, a += step * pa) {
          const float hv = *a;
          const int64_t env = tile_base + r;
          float* hb = bf.actor_history + (env * H) * per + c;
          float* po = pol + (int64_t)r * P + Pcur + c;
          if (bf.just_reset[env]) {
            for (int i = 0; i < H; ++i) {
              hb[(int64_t)i * per] = hv;
              __builtin_nontemporal_store(hv, po + (int64_t)i * per);
            }
          } else {
            for (int i = H - 2; i >= 0; --i) {
              const float x = hb[(int64_t)i * per];
              hb[(int64_t)(i + 1) * per] = x;
              __builtin_nontemporal_store(x, po + (int64_t)(i + 1) * per);
            }
            hb[0] = hv;
            __builtin_nontemporal_store(hv, po);
          }
        }
      };
      if (per >= kBlock) {
        for (int c = tid; c < per; c += kBlock) column(c, 0, 1);
      } else {
        const int G = kBlock / per, g = row_of(tid, 1.0f / (float)per);
        if (g < G) column(tid - g * per, g, G);
      }
      __syncthreads();  // every lane has read its rows' flags
      if (tid < T) bf.just_reset[tile_base + tid] = 0;
    }
  }
}

template <int kTileEnvs>
__global__ __launch_bounds__(kBlock) void env_step_kernel(EnvPlan p, AmpSimState st, AmpEnvBuffers bf, int64_t N,
                                                          unsigned block0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  env_step_body<kTileEnvs>(p, st, bf, N, (int64_t)(block0 + blockIdx.x), smem);
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
  int32_t wide;          // 256-sample workgroups (collect_reference_wide_body) instead of 64-sample ones
  unsigned first_blocks; // expert workgroups that run BEFORE the env tiles (the rest run after them)
};
// which workgroup is which in a fused launch: [0, first) expert | [first, first + env_blocks) env tiles | rest expert
__device__ __forceinline__ bool fused_is_env(unsigned b, unsigned env_blocks, const ExpertArgs& x, int64_t& idx) {
  if (b < x.first_blocks) { idx = b; return false; }
  if (b < x.first_blocks + env_blocks) { idx = b - x.first_blocks; return true; }
  idx = b - env_blocks;
  return false;
}
__device__ __forceinline__ void fused_expert(const ExpertArgs& x, int64_t idx, float* smem) {
  if (x.wide) collect_reference_wide_body(x.v, x.times, x.ids, x.n, x.K, x.out, idx, smem);
  else collect_reference_body(x.v, x.times, x.ids, x.n, x.K, x.out, nullptr, nullptr, idx, smem);
}
template <int kTileEnvs>
__global__ __launch_bounds__(kBlock) void env_step_reference_kernel(EnvPlan p, AmpSimState st, AmpEnvBuffers bf, int64_t N,
                                                                    unsigned env_blocks, ExpertArgs x) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int64_t idx;
  if (fused_is_env(blockIdx.x, env_blocks, x, idx)) env_step_body<kTileEnvs>(p, st, bf, N, idx, smem);
  else fused_expert(x, idx, smem);
}
template <int T>
__global__ __launch_bounds__(kBlock) void env_step_dma_kernel(EnvPlan p, AmpSimState st, AmpEnvBuffers bf, int64_t N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  env_step_dma_pass<T>(p, st, bf, N, (int64_t)blockIdx.x, smem);
}
template <int T>
__global__ __launch_bounds__(kBlock) void env_step_dma_reference_kernel(EnvPlan p, AmpSimState st, AmpEnvBuffers bf, int64_t N,
                                                                        unsigned env_blocks, ExpertArgs x) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int64_t idx;
  if (fused_is_env(blockIdx.x, env_blocks, x, idx)) env_step_dma_pass<T>(p, st, bf, N, idx, smem);
  else fused_expert(x, idx, smem);
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

static int pick_tile(const EnvPlan& p, int64_t num_envs) {
  // envs per workgroup.  Generic body, measured (MI355X, G1 K=2): 65 536 envs 72 / 78 / 91 us with 64 / 32 / 16-env
  // tiles; 16 384 envs 39 / 27 / 23 us.  DMA body inside the step: 16 384 envs 21.5 us with 16-env tiles, 17.7 us with 32
  // (per-workgroup setup and per-env work amortise over twice the rows); 8 192 envs 14.9 vs 14.1 us (equal: one workgroup
  // per CU either way).  Shards of 16 384 envs and more use 32, smaller ones are spread over the whole chip with 16
  // (latency-bound there); the tile is halved (down to 8) until the DMA body's [tile, K*D] LDS image fits 64 KB (K = 10
  // with 16-env tiles in 70 KB, two workgroups per CU: 216 vs 192 us per launch at 65 536 envs -- 8-env tiles stay).
  int tile = num_envs >= 16 * 1024 ? 32 : 16;
  while (tile > 8 && sizeof(float) * (size_t)env_dma_lds_floats(tile, p.K * p.D, p.n_dof, true) > 64 * 1024) tile >>= 1;
  return tile;
}

int32_t amp_env_step_tile_envs(const AmpEnvCfg* cfg, int64_t num_envs) {
  EnvPlan p;
  if (!cfg || make_plan(cfg, 0, &p) != AMP_OK) return -1;
  return pick_tile(p, num_envs);
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
  const int tile = pick_tile(p, N);
  const size_t lds = sizeof(float) * ((size_t)tile * p.D + 2 * (size_t)tile * p.dof_pad + 4 * tile) + sizeof(int) * tile +
                     sizeof(float) * (size_t)(per_env_limits ? tile : 1) * (2 * p.n_dof + 1) +
                     sizeof(float) * 2 * (size_t)p.K * p.D;  // + the scaler statistics of the fused discriminator input
  AMP_REQUIRE(lds <= 64 * 1024, "amp_env_step: observation tile needs %zu B of LDS (> 64 KiB)", lds);
  const unsigned grid = (unsigned)((N + tile - 1) / tile);
  const bool with_expert = expert && expert->n > 0;
  // expert tiles of the fused launch: 256-sample workgroups once there are enough samples to give every CU one of them,
  // 64-sample workgroups below that (small shards are latency-bound: more, shorter workgroups).  They take the FIRST
  // block indices: their dependent gather chains then run under the env tiles' streaming instead of after it (measured in
  // the step: 65 536 envs 331 -> 329 us, 8 192 envs 62.7 -> 61.3 us, humanoid 32 768 envs 29.8 -> 27.8 us for the launch).
  ExpertArgs xa{};
  if (with_expert) {
    xa = *expert;
    const int64_t total = expert->n * expert->K;
    xa.wide = total >= 256 * (int64_t)kExpertWide && expert_wide_lds(expert->v.D) <= 64 * 1024;
#ifdef AMP_EXPERT_NO_WIDE  // diagnostic builds only: the 64-sample expert body (22 KB of LDS) -> four env workgroups per CU
    xa.wide = 0;
#endif
    expert = &xa;
  }
  const size_t lds_x = with_expert ? (xa.wide ? expert_wide_lds(expert->v.D) : expert_lds(expert->v.D)) : 0;
  const int x_tile = xa.wide ? kExpertWide : kExpertTile;
  const unsigned grid_x = with_expert ? (unsigned)((expert->n * expert->K + x_tile - 1) / x_tile) : 0;
  xa.first_blocks = grid_x;
  AMP_REQUIRE(lds_x <= 64 * 1024, "amp_env_step_with_reference: tile needs %zu B of LDS (> 64 KiB)", lds_x);
  const char* label = with_expert ? "env_step_reference_kernel" : "env_step_kernel";

  // The hot-path configuration runs the DMA tile body on every whole tile (see env_step_dma_pass); a ragged last
  // tile, and every other configuration, runs the generic body.
  auto aligned = [](const void* ptr, uintptr_t a) { return (reinterpret_cast<uintptr_t>(ptr) & (a - 1)) == 0; };
  auto rows16 = [&](const float* ptr, int64_t stride) { return stride == p.n_dof && aligned(ptr, 16); };
  // DMA tile body: any phase subset, whole tiles, K*D even; the alignment conditions of a phase's inputs
  // and outputs apply only when that phase is asked for
  const int KD = p.K * p.D;
  const bool obs = phases & AMP_PHASE_OBS;
  const size_t lds_dma = sizeof(float) * (size_t)env_dma_lds_floats(tile, KD, p.n_dof, per_env_limits, obs);
  bool dma = (KD & 1) == 0 && N >= tile && lds_dma <= 64 * 1024;
  if (dma && g1_rew) dma = rows16(st->actions, st->actions_stride) && rows16(st->joint_acc, st->joint_acc_stride);
  if (dma && obs) {
    dma = aligned(bf->amp_obs_buffer, 16) && aligned(bf->policy_obs, 8);
    if (dma && p.use_last_actions) dma = aligned(st->last_actions, 16) && (!p.use_command || aligned(st->command, 16));
    if (dma && bf->disc_input) dma = aligned(bf->disc_input, 8) && (bf->disc_input_stride & 1) == 0;
  }
  auto generic_tile = [&](unsigned g, unsigned block0) {
    if (tile == 32) env_step_kernel<32><<<g, kBlock, lds, (hipStream_t)stream>>>(p, *st, *bf, N, block0);
    else if (tile == 16) env_step_kernel<16><<<g, kBlock, lds, (hipStream_t)stream>>>(p, *st, *bf, N, block0);
    else env_step_kernel<8><<<g, kBlock, lds, (hipStream_t)stream>>>(p, *st, *bf, N, block0);
  };
  if (dma) {
    const unsigned full = (unsigned)(N / tile);
    const size_t lds_f = lds_dma > lds_x ? lds_dma : lds_x;
    {
      amp::TraceScope trace__(label, (hipStream_t)stream);
      if (with_expert) {
        if (tile == 32) env_step_dma_reference_kernel<32><<<full + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, full, *expert);
        else if (tile == 16) env_step_dma_reference_kernel<16><<<full + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, full, *expert);
        else env_step_dma_reference_kernel<8><<<full + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, full, *expert);
      } else {
        if (tile == 32) env_step_dma_kernel<32><<<full, kBlock, lds_dma, (hipStream_t)stream>>>(p, *st, *bf, N);
        else if (tile == 16) env_step_dma_kernel<16><<<full, kBlock, lds_dma, (hipStream_t)stream>>>(p, *st, *bf, N);
        else env_step_dma_kernel<8><<<full, kBlock, lds_dma, (hipStream_t)stream>>>(p, *st, *bf, N);
      }
    }
    if (full != grid) generic_tile(1, full);  // ragged last tile
    return launch_status(label);
  }
  if (with_expert) {
    const size_t lds_f = lds > lds_x ? lds : lds_x;
    amp::TraceScope trace__(label, (hipStream_t)stream);
    if (tile == 32) env_step_reference_kernel<32><<<grid + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, grid, *expert);
    else if (tile == 16) env_step_reference_kernel<16><<<grid + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, grid, *expert);
    else env_step_reference_kernel<8><<<grid + grid_x, kBlock, lds_f, (hipStream_t)stream>>>(p, *st, *bf, N, grid, *expert);
    return launch_status(label);
  }
  { amp::TraceScope trace__(label, (hipStream_t)stream);
    generic_tile(grid, 0u);
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
  AMP_REQUIRE(n_samples * (int64_t)K < (int64_t)1 << 31, "amp_env_step_with_reference: more than 2^31 - 1 expert samples");
  ExpertArgs x{motion->v, times_dev, motion_ids_dev, n_samples, K, expert_out_dev, 0, 0u};
  return env_step_launch(cfg, st, bf, N, phases, &x, stream);
}

}  // extern "C"
