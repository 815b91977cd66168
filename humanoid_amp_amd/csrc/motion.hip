// Motion table kernels: frame/blend index (fp64), LERP x5 + SLERP sampling, fused expert AMP
// observations, reference-state reset.  gfx950 only; built with -ffp-contract=off (SURVEY.md section 7:
// sqrt(1 - c*c) must not become an fma, the quaternion dot must keep its rounding).
//
// Work decomposition (all kernels): a workgroup owns a TILE of 256 consecutive samples.
//   phase A  one sample per lane: fp64 frame/blend index math (+ the per-sample quaternion work) -> LDS
//   phase B  the tile's output is one contiguous run of floats; lanes walk it flat, so every store is a
//            full-wave coalesced 256-B write, and the two bracketing table rows of a sample are read as
//            contiguous row segments (the tables are <= 1.4 MB: L2 / Infinity-Cache resident).
#include "amp_common.hpp"

#include "motion_kernels.hpp"
#include "compact_kernels.hpp"
#include "command_kernels.hpp"

namespace amp {

__global__ __launch_bounds__(kBlock) void frame_blend_kernel(ClipMeta m, const double* __restrict__ times,
                                                             const int64_t* __restrict__ ids, int64_t n,
                                                             int64_t* __restrict__ o0, int64_t* __restrict__ o1,
                                                             double* __restrict__ ob) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int64_t a, b;
  double w;
  frame_blend_ref(m, times[i], ids ? ids[i] : 0, a, b, w);
  o0[i] = a;
  o1[i] = b;
  ob[i] = w;
}

// hot[f] = [dof_pos[perm] | dof_vel[perm] | ref pos 3 | ref quat 4 | ref lin 3 | ref ang 3 | key pos 3*n_key]
__global__ __launch_bounds__(kBlock) void build_hot_kernel(MotionView v, int64_t n_frames, const int32_t* __restrict__ perm,
                                                           int32_t ref, const int32_t* __restrict__ keys,
                                                           float* __restrict__ hot) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int D = v.D, HP = v.HP;
  if (e >= n_frames * HP) return;
  const int64_t f = e / HP;
  const int j = (int)(e - f * HP);
  if (j >= D) {
    hot[e] = 0.0f;  // row padding
    return;
  }
  const int nd = v.n_dof, B = v.n_bodies;
  float val;
  if (j < nd) {
    val = v.dof_pos[f * nd + perm[j]];
  } else if (j < 2 * nd) {
    val = v.dof_vel[f * nd + perm[j - nd]];
  } else {
    const int c = j - 2 * nd;
    if (c < 3) val = v.body_pos[(f * B + ref) * 3 + c];
    else if (c < 7) val = v.body_rot[(f * B + ref) * 4 + (c - 3)];
    else if (c < 10) val = v.body_lin[(f * B + ref) * 3 + (c - 7)];
    else if (c < 13) val = v.body_ang[(f * B + ref) * 3 + (c - 10)];
    else {
      const int kk = (c - 13) / 3, ax = (c - 13) % 3;
      val = v.body_pos[(f * B + keys[kk]) * 3 + ax];
    }
  }
  hot[e] = val;
}

// samp[f] = [dof_pos | dof_vel | body_pos | body_lin | body_ang | body_rot], segments padded to 4 floats (zeros)
__global__ __launch_bounds__(kBlock) void build_sample_table_kernel(MotionView v, int64_t n_frames, float* __restrict__ samp) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int SP = v.SP, nd = v.n_dof, B = v.n_bodies, dw = samp_dof_w(nd), bw = samp_body_w(B);
  if (e >= n_frames * SP) return;
  const int64_t f = e / SP;
  int j = (int)(e - f * SP);
  float val = 0.0f;
  if (j < dw) { if (j < nd) val = v.dof_pos[f * nd + j]; }
  else if ((j -= dw) < dw) { if (j < nd) val = v.dof_vel[f * nd + j]; }
  else if ((j -= dw) < bw) { if (j < 3 * B) val = v.body_pos[f * 3 * B + j]; }
  else if ((j -= bw) < bw) { if (j < 3 * B) val = v.body_lin[f * 3 * B + j]; }
  else if ((j -= bw) < bw) { if (j < 3 * B) val = v.body_ang[f * 3 * B + j]; }
  else { j -= bw; val = v.body_rot[f * 4 * B + j]; }
  samp[e] = val;
}

// ------------------------------------------------------------------------------------------------
// MotionLoader.sample: 6 outputs (motions/motion_loader.py:331-390)
//
// A workgroup owns T consecutive samples (64 / 32 / 16 / 8: the largest whose six output tiles fit 56 KB of LDS).
//   phase A  one sample per lane: fp64 frame / blend index -> LDS slot (i0, i1, blend)
//   phase B  a lane owns ONE quad of columns of the padded sample table (which output it feeds, where its four columns land
//            in that output's LDS tile, which of them are padding: per-lane constants) and walks the tile's samples: two 16-B
//            gathers + four LERPs per item, or -- for the quads of the rotation segment -- one SLERP; no per-element row /
//            column arithmetic (the first version walked every output flat with 4-B gathers, an LDS slot read and a
//            (sample, column) carry per element: 2.0 TB/s of output at 131 072 samples of G1_walk)
//   phase C  each output's tile is one contiguous run of T * row floats in HBM: streamed out of LDS with 16-B stores
// Same arithmetic per element as before (lerp_ref / slerp_ref on the same operands): bit-identical.
// ------------------------------------------------------------------------------------------------
struct SampleSlot {
  int32_t i0, i1;
  float blend;
};

__host__ __device__ inline int sample_out_floats(int n_dof, int n_bodies) { return 2 * n_dof + 13 * n_bodies; }
static inline int sample_tile(int n_dof, int n_bodies) {
  int T = 64;
  while (T > 8 && (size_t)T * sample_out_floats(n_dof, n_bodies) * sizeof(float) > 56 * 1024) T >>= 1;
  return T;
}
static inline size_t sample_lds(int T, int n_dof, int n_bodies) {
  return sizeof(float) * ((size_t)T * sample_out_floats(n_dof, n_bodies) + 4) + sizeof(SampleSlot) * T;
}

struct SampleOuts {
  float* o[6];  // dof_pos, dof_vel, body_pos, body_lin, body_ang, body_rot (the sample table's segment order)
};

__global__ __launch_bounds__(kBlock) void sample_kernel(MotionView v, const double* __restrict__ times,
                                                        const int64_t* __restrict__ ids, int64_t n, int T, SampleOuts out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nd = v.n_dof, B = v.n_bodies, dw = samp_dof_w(nd), bw = samp_body_w(B);
  const int tid = threadIdx.x;
  // LDS: the six output tiles back to back ([T][row] each, 16-B aligned starts: T * row is a multiple of 4), a trash quad, slots
  const int row[6] = {nd, nd, 3 * B, 3 * B, 3 * B, 4 * B};
  const int segw[5] = {dw, dw, bw, bw, bw};
  int img[6];
  {
    int o = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) { img[a] = o; o += T * row[a]; }
  }
  const int total = T * sample_out_floats(nd, B);
  SampleSlot* const slots = reinterpret_cast<SampleSlot*>(smem + total + 4);
  const int64_t tile_base = (int64_t)blockIdx.x * T;
  const int n_tile = (int)((n - tile_base) < T ? (n - tile_base) : T);
  // ---- phase A ---------------------------------------------------------------------------------------------------
  if (tid < n_tile) {
    const int64_t i = tile_base + tid;
    int64_t a, b;
    double w;
    frame_blend_ref(v.clips, times[i], ids ? ids[i] : 0, a, b, w);
    slots[tid] = SampleSlot{(int32_t)a, (int32_t)b, (float)w};
  }
  __syncthreads();
  const int SQ = v.SP >> 2;                       // quads per table row
  const int QL = (2 * dw + 3 * bw) >> 2;          // quads of the five LERP segments; the B rotation quads follow
  const f4* __restrict__ samp4 = reinterpret_cast<const f4*>(v.samp);
  // ---- phase B, LERP segments: passes of up to 256 quads, G = 256 / quads sample groups side by side ---------------
  for (int q0 = 0; q0 < QL; q0 += kBlock) {
    const int Qp = QL - q0 < kBlock ? QL - q0 : kBlock;
    const int G = kBlock / Qp, g = (int)(((float)tid + 0.5f) * (1.0f / (float)Qp));
    if (g < G) {
      const int q = q0 + tid - g * Qp;
      // which segment the quad belongs to and its first column inside it
      int seg = 0, c0 = 4 * q;
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (seg == a && c0 >= segw[a]) { c0 -= segw[a]; seg = a + 1; }
      int rw = row[0], im = img[0];
      const float* optr = out.o[0];
#pragma unroll
      for (int a = 1; a < 5; ++a)
        if (seg == a) { rw = row[a]; im = img[a]; optr = out.o[a]; }
      int woff[4], wstep[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool keep = optr != nullptr && c0 + c < rw;   // padding columns and skipped outputs go to the trash quad
        woff[c] = keep ? im + g * rw + c0 + c : total + c;
        wstep[c] = keep ? G * rw : 0;
      }
      float* w0 = smem + woff[0]; float* w1 = smem + woff[1]; float* w2 = smem + woff[2]; float* w3 = smem + woff[3];
      // four samples per trip: their eight 16-B gathers are in flight together (the walk is latency-bound: L2 round trips)
      constexpr int U = 4;
      for (int s = g; s < n_tile; s += U * G) {
        SampleSlot k[U];
        f4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int su = s + u * G < n_tile ? s + u * G : n_tile - 1;  // clamped: a repeated (cached) gather, result dropped
          k[u] = slots[su];
          a[u] = samp4[(int64_t)k[u].i0 * SQ + q];
          b[u] = samp4[(int64_t)k[u].i1 * SQ + q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (s + u * G < n_tile) {
            *w0 = lerp_ref(a[u][0], b[u][0], k[u].blend);
            *w1 = lerp_ref(a[u][1], b[u][1], k[u].blend);
            *w2 = lerp_ref(a[u][2], b[u][2], k[u].blend);
            *w3 = lerp_ref(a[u][3], b[u][3], k[u].blend);
            w0 += wstep[0]; w1 += wstep[1]; w2 += wstep[2]; w3 += wstep[3];
          }
      }
    }
  }
  // ---- phase B, rotations: one quaternion (= one quad) per item, wxyz SLERP (motion_loader.py:217-279) --------------
  if (out.o[5]) {
    for (int b0 = 0; b0 < B; b0 += kBlock) {
      const int Qp = B - b0 < kBlock ? B - b0 : kBlock;
      const int G = kBlock / Qp, g = (int)(((float)tid + 0.5f) * (1.0f / (float)Qp));
      if (g < G) {
        const int b = b0 + tid - g * Qp;
        const int q = QL + b;
        f4* w = reinterpret_cast<f4*>(smem + img[5]) + g * B + b;   // the rotation tile starts 16-B aligned, rows of B quads
        const SampleSlot* sl = slots + g;
        for (int s = g; s < n_tile; s += G, sl += G, w += G * B) {
          const SampleSlot k = *sl;
          const f4 a = samp4[(int64_t)k.i0 * SQ + q];
          const f4 c = samp4[(int64_t)k.i1 * SQ + q];
          const Quat o = slerp_ref(Quat{a[0], a[1], a[2], a[3]}, Quat{c[0], c[1], c[2], c[3]}, k.blend);
          *w = f4{o.w, o.x, o.y, o.z};
        }
      }
    }
  }
  __syncthreads();
  // ---- phase C: every output's tile is one contiguous run ---------------------------------------------------------------
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float* dst = out.o[a];
    if (!dst) continue;
    dst += tile_base * row[a];
    const float* src = smem + img[a];
    const int count = n_tile * row[a];
    if ((((uintptr_t)dst) & 15) == 0) {
      const f4* src4 = reinterpret_cast<const f4*>(src);
      f4* dst4 = reinterpret_cast<f4*>(dst);
      for (int e = tid; e < (count >> 2); e += kBlock) dst4[e] = src4[e];
      for (int e = (count & ~3) + tid; e < count; e += kBlock) dst[e] = src[e];
    } else {
      for (int e = tid; e < count; e += kBlock) dst[e] = src[e];
    }
  }
}

// collect_reference_motions fused with compute_obs: the tile body lives in motion_kernels.hpp
__global__ __launch_bounds__(kBlock) void collect_reference_kernel(MotionView v, const double* __restrict__ times,
                                                                   const int64_t* __restrict__ ids, int64_t n, int K,
                                                                   float* __restrict__ out,
                                                                   const int64_t* __restrict__ dst_rows,
                                                                   const int64_t* __restrict__ n_dev) {
  extern __shared__ __attribute__((aligned(16))) float s_img[];
  collect_reference_body(v, times, ids, n, K, out, dst_rows, n_dev, (int64_t)blockIdx.x, s_img);
}

// the hot-path form (contiguous rows, < 2^31 samples): 256-sample workgroups, see collect_reference_wide_body
__global__ __launch_bounds__(kBlock) void collect_reference_wide_kernel(MotionView v, const double* __restrict__ times,
                                                                        const int64_t* __restrict__ ids, int64_t n, int K,
                                                                        float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_img[];
  collect_reference_wide_body(v, times, ids, n, K, out, (int64_t)blockIdx.x, s_img);
}

// ------------------------------------------------------------------------------------------------
// reference-state init of reset envs (g1_amp_env.py:385-411)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void reset_state_kernel(MotionView v, const double* __restrict__ times,
                                                             const int64_t* __restrict__ ids,
                                                             const int64_t* __restrict__ env_ids, int64_t n,
                                                             const float* __restrict__ origins, float z_lift,
                                                             float* __restrict__ root, float* __restrict__ o_dp,
                                                             float* __restrict__ o_dv, const int64_t* __restrict__ n_dev) {
  __shared__ SampleSlot slots[kBlock];
  const int nd = v.n_dof;
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  const int64_t tile_base = (int64_t)blockIdx.x * kBlock;
  if (tile_base >= n) return;
  const int n_tile = (int)((n - tile_base) < kBlock ? (n - tile_base) : kBlock);
  const float* __restrict__ hot = v.hot;
  if (threadIdx.x < n_tile) {
    const int64_t i = tile_base + threadIdx.x;
    int64_t a, b;
    double w;
    frame_blend_ref(v.clips, times[i], ids ? ids[i] : 0, a, b, w);
    const float bl = (float)w;
    slots[threadIdx.x] = SampleSlot{(int32_t)a, (int32_t)b, bl};
    if (root) {
      const float* r0 = hot + a * v.HP + 2 * nd;
      const float* r1 = hot + b * v.HP + 2 * nd;
      const int64_t env = env_ids ? env_ids[i] : i;
      float* o = root + i * 13;
      const float* og = origins ? origins + env * 3 : nullptr;
      const float px = lerp_ref(r0[0], r1[0], bl), py = lerp_ref(r0[1], r1[1], bl), pz = lerp_ref(r0[2], r1[2], bl);
      o[0] = og ? px + og[0] : px;
      o[1] = og ? py + og[1] : py;
      o[2] = (og ? pz + og[2] : pz) + z_lift;
      const Quat q = slerp_ref(Quat{r0[3], r0[4], r0[5], r0[6]}, Quat{r1[3], r1[4], r1[5], r1[6]}, bl);
      o[3] = q.w; o[4] = q.x; o[5] = q.y; o[6] = q.z;
      for (int c = 7; c < 13; ++c) o[c] = lerp_ref(r0[c], r1[c], bl);
    }
  }
  __syncthreads();
  // DoF positions / velocities are hot columns [0, nd) and [nd, 2nd), already in robot order
  for (int which = 0; which < 2; ++which) {
    float* out = which ? o_dv : o_dp;
    if (!out) continue;
    const int off = which * nd;
    const int step_s = kBlock / nd, step_j = kBlock % nd;
    int s = threadIdx.x / nd, j = threadIdx.x % nd;
    float* o = out + tile_base * nd;
    for (int e = threadIdx.x; e < n_tile * nd; e += kBlock) {
      const SampleSlot sl = slots[s];
      o[e] = lerp_ref(hot[(int64_t)sl.i0 * v.HP + off + j], hot[(int64_t)sl.i1 * v.HP + off + j], sl.blend);
      s += step_s;
      j += step_j;
      if (j >= nd) {
        j -= nd;
        s += 1;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// device-side sample_times (replaces the host numpy RNG of motion_loader.py:309-329 on the reset path).
// Counter-based: Philox4x32-10, key = seed, counter = (index, step), so the draw of an env does not depend on
// how many other envs reset in the same step.  Parity with the reference is distributional only (it uses numpy's
// global MT19937); bit-exactness is defined against oracle/rng.py.
// ------------------------------------------------------------------------------------------------
// the (clip, time) draw of one reset env: key = seed, counter = (global env id, step)
__device__ __forceinline__ void draw_clip_time(const ClipMeta& m, uint64_t seed, uint64_t step, int start, uint64_t ctr,
                                               int64_t& clip, double& t) {
  uint32_t r[4];
  philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), r);
  clip = (int64_t)(((uint64_t)r[0] * (uint64_t)m.n_clips) >> 32);  // uniform in [0, n_clips)
  // 53-bit uniform in [0, 1) from two words, numpy's random_sample construction
  const double u = ((double)(r[1] >> 5) * 67108864.0 + (double)(r[2] >> 6)) / 9007199254740992.0;
  t = start ? 0.0 : u * m.dur[clip];
}

// per-env clears of a reset env (DirectRLEnv._reset_idx: episode_length_buf[env_ids] = 0; g1_amp_env.py:352-358:
// _just_reset_mask[env_ids] = True); the last_actions row is cleared by the caller's row loop or here (serial)
struct ResetClears {
  int64_t* episode_length;
  float* last_actions;
  uint8_t* just_reset;
  int32_t n_actions;
};

__global__ __launch_bounds__(kBlock) void sample_times_kernel(ClipMeta m, uint64_t seed, uint64_t step, int start,
                                                              const int64_t* __restrict__ index, int64_t n,
                                                              const int64_t* __restrict__ n_dev,
                                                              int64_t* __restrict__ o_ids, double* __restrict__ o_t,
                                                              int64_t* __restrict__ env_ids_out,
                                                              float* __restrict__ env_t_out, int64_t ctr_offset, ResetClears cl,
                                                              const uint64_t* __restrict__ step_dev) {
  if (step_dev) step += *step_dev;
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint64_t idx = (uint64_t)(index ? index[i] : i);
  int64_t clip;
  double t;
  // counter = global env id of a shard's local env: the draw is shard-invariant
  draw_clip_time(m, seed, step, start, idx + (uint64_t)ctr_offset, clip, t);
  o_ids[i] = clip;
  o_t[i] = t;
  // per-env mirrors (self.motion_ids[env_ids] = ..., self.motion_start_times[env_ids] = ..., g1_amp_env.py:377-382)
  if (env_ids_out) env_ids_out[idx] = clip;
  if (env_t_out) env_t_out[idx] = (float)t;
  if (cl.episode_length) cl.episode_length[idx] = 0;
  if (cl.just_reset) cl.just_reset[idx] = 1;
  if (cl.last_actions)
    for (int j = 0; j < cl.n_actions; ++j) cl.last_actions[idx * cl.n_actions + j] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// The whole device-side reset as ONE launch (amp_reset_compact_apply): reset-id compaction + everything amp_reset_apply
// does + the reset-side command resample + the per-env clears (+ optionally the reward-log means of the step, on extra
// workgroups).  A workgroup owns 256 consecutive envs: compact_rank_body gives every reset lane its slot in the ascending
// id list (and writes ids / count exactly as the stand-alone compaction does); the lane draws (clip, t) and queues its env
// in an LDS list; the workgroup's reset envs (0-3 of 256 in steady state) are then finished by collect_reference_body over
// that list (times / clips / destination rows read from LDS): phase A's gathers also yield the root state, the k = 0
// frame's first 2 n_dof columns are the DoF rows.  A chain of four dependent memory round trips (counts -> clip table ->
// frame gathers -> column gathers) instead of the eight of the separate launches' bodies run back to back.
// Same device functions on the same inputs per env: bit-identical to the separate launches.
// ------------------------------------------------------------------------------------------------
// one term's mean over the envs on a 256-lane workgroup: fp64 accumulation, fixed order (eight 16-B loads in flight per lane)
__device__ __forceinline__ void reward_mean_block(const float* __restrict__ row, int64_t N, float* __restrict__ mean_out) {
  __shared__ double red[kBlock];
  const int tid = threadIdx.x;
  double s = 0.0;
  if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
    const f4* row4 = reinterpret_cast<const f4*>(row);
    const int64_t n4 = N >> 2;
    for (int64_t i = tid; i < n4; i += 8 * kBlock) {
      f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = i + u * kBlock < n4 ? row4[i + u * kBlock] : f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < 8; ++u) s += ((double)v[u][0] + (double)v[u][1]) + ((double)v[u][2] + (double)v[u][3]);
    }
    for (int64_t i = (n4 << 2) + tid; i < N; i += kBlock) s += (double)row[i];
  } else {
    for (int64_t i = tid; i < N; i += kBlock) s += (double)row[i];
  }
  red[tid] = s;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) *mean_out = (float)(red[0] / (double)N);
}

__global__ __launch_bounds__(kBlock) void reset_compact_apply_kernel(MotionView v, AmpCompactArgs c, AmpResetArgs a,
                                                                     AmpCommandArgs cmd, int has_cmd, int64_t n_tiles, int sub,
                                                                     int64_t n_counts, unsigned compact_blocks,
                                                                     AmpRewardLogArgs lg) {
  extern __shared__ __attribute__((aligned(16))) float s_img[];  // expert_lds(D): collect_reference_body's tile
  if (blockIdx.x >= compact_blocks) {  // the step's reward-log means ride on this launch: one workgroup per term
    const int term = (int)(blockIdx.x - compact_blocks);
    reward_mean_block(lg.reward_terms + (int64_t)term * c.num_envs, c.num_envs, lg.means + term);
    return;
  }
  __shared__ double s_t[kBlock];
  __shared__ int64_t s_clip[kBlock], s_env[kBlock], s_slot[kBlock];
  __shared__ int s_wcnt[kBlock / kWave];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the step counter's hand-back inside a captured step (AmpPrePhysicsArgs.step_in / step_out): nothing in this launch reads
  // *step_dev_out, and *step_dev is only read
  if (a.step_dev_out && blockIdx.x == 0 && tid == 0) *a.step_dev_out = *a.step_dev;
  int64_t env;
  const int64_t slot = compact_rank_body((int64_t)blockIdx.x, c.mask, c.tile_counts, c.num_envs, n_tiles, sub, n_counts, c.ids,
                                         c.count, env);
  const unsigned long long b = __ballot(slot >= 0);
  if (lane == 0) s_wcnt[wave] = __popcll(b);
  __syncthreads();
  int wbase = 0, cnt = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) {
    wbase += w < wave ? s_wcnt[w] : 0;
    cnt += s_wcnt[w];
  }
  if (cnt == 0) return;  // uniform: nothing to reset among this workgroup's envs
  if (a.mode == AMP_RESET_DEFAULT) {
    // reset_strategy "default" (g1_amp_env.py:338-339, 362-369): default root state (+ env origin) and default joint state for
    // the provider's write, the per-env clears of _reset_idx (:352-358); no draw, no expert frames, no command resample
    if (slot >= 0) {
      const int li = wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
      s_env[li] = env;
      s_slot[li] = slot;
      if (a.episode_length) a.episode_length[env] = 0;
      if (a.just_reset) a.just_reset[env] = 1;
    }
    __syncthreads();
    const int nd = v.n_dof, width = 13 + 2 * nd;
    for (int e = tid; e < cnt * width; e += kBlock) {
      const int li = e / width, j = e - li * width;
      const int64_t en = s_env[li], sl = s_slot[li];
      if (j < 13) {
        float x = a.default_root_state[en * 13 + j];
        if (j < 3 && a.env_origins) x += a.env_origins[en * 3 + j];
        if (a.root_state) a.root_state[sl * 13 + j] = x;
      } else if (j < 13 + nd) {
        if (a.dof_pos) a.dof_pos[sl * nd + (j - 13)] = a.default_joint_pos[en * nd + (j - 13)];
      } else {
        if (a.dof_vel) a.dof_vel[sl * nd + (j - 13 - nd)] = a.default_joint_vel[en * nd + (j - 13 - nd)];
      }
    }
    if (a.last_actions)
      for (int e = tid; e < cnt * a.n_actions; e += kBlock) {
        const int li = e / a.n_actions;
        a.last_actions[s_env[li] * a.n_actions + (e - li * a.n_actions)] = 0.0f;
      }
    return;
  }
  if (slot >= 0) {
    const int li = wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
    int64_t clip;
    double t;
    draw_clip_time(v.clips, a.seed, a.step + (a.step_dev ? *a.step_dev : 0ull), a.start, (uint64_t)env + (uint64_t)a.env_offset, clip, t);
    a.motion_ids[slot] = clip;
    a.motion_times[slot] = t;
    if (a.env_motion_ids) a.env_motion_ids[env] = clip;
    if (a.env_motion_start_times) a.env_motion_start_times[env] = (float)t;
    s_t[li] = t;
    s_clip[li] = clip;
    s_env[li] = env;
    s_slot[li] = slot;
    if (a.episode_length) a.episode_length[env] = 0;
    if (a.just_reset) a.just_reset[env] = 1;
    if (has_cmd) command_reset_env(cmd, env);
  }
  __syncthreads();
  if (a.last_actions)
    for (int e = tid; e < cnt * a.n_actions; e += kBlock) {
      const int li = e / a.n_actions;
      a.last_actions[s_env[li] * a.n_actions + (e - li * a.n_actions)] = 0.0f;
    }
  // K expert frames per reset env into amp_observation_buffer[env] (g1_amp_env.py:414-419); their phase A also writes the
  // root state, their k = 0 frames the DoF rows
  const ResetRows rr{a.root_state, a.dof_pos, a.dof_vel, s_slot, a.env_origins, a.z_lift};
  for (int chunk = 0; chunk * kExpertTile < cnt * a.K; ++chunk) {
    collect_reference_body<true>(v, s_t, s_clip, cnt, a.K, a.amp_obs_buffer, s_env, nullptr, chunk, s_img, &rr);
    __syncthreads();
  }
}

static inline unsigned grid_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace amp

using namespace amp;

extern "C" {

int amp_motion_create(const AmpMotionDesc* d, AmpMotion** out) {
  AMP_REQUIRE(d && out, "amp_motion_create: null argument");
  AMP_REQUIRE(d->n_clips >= 1 && d->n_dof >= 1 && d->n_bodies >= 1, "amp_motion_create: empty shape");
  AMP_REQUIRE(d->clip_frames, "amp_motion_create: clip_frames is null");
  AMP_REQUIRE(d->dt > 0.0, "amp_motion_create: dt must be positive");
  AMP_REQUIRE(d->dof_positions && d->dof_velocities && d->body_positions && d->body_rotations &&
                  d->body_linear_velocities && d->body_angular_velocities,
              "amp_motion_create: a motion table pointer is null");
  int64_t total = 0;
  for (int c = 0; c < d->n_clips; ++c) {
    AMP_REQUIRE(d->clip_frames[c] >= 2, "amp_motion_create: clip %d has %lld frame(s); at least 2 are needed", c,
                (long long)d->clip_frames[c]);
    total += d->clip_frames[c];
  }
  AMP_REQUIRE(total == d->n_frames, "amp_motion_create: clip_frames sum to %lld but n_frames is %lld", (long long)total,
              (long long)d->n_frames);
  AMP_REQUIRE(total < (int64_t)1 << 31, "amp_motion_create: more than 2^31 frames");
  AmpMotion* h = new (std::nothrow) AmpMotion();
  AMP_REQUIRE(h, "amp_motion_create: out of host memory");
  *h = AmpMotion{};
  AMP_HIP(hipGetDevice(&h->device));
  h->n_frames = total;
  int64_t* first = new int64_t[d->n_clips];
  int64_t* span = new int64_t[d->n_clips];
  double* dur = new double[d->n_clips];
  int64_t cur = 0;
  for (int c = 0; c < d->n_clips; ++c) {
    first[c] = cur;
    span[c] = d->clip_frames[c] - 1;
    dur[c] = d->dt * (double)(d->clip_frames[c] - 1);  // motions/motion_loader.py:135
    cur += d->clip_frames[c];
  }
  hipError_t e = hipMalloc(&h->d_first, sizeof(int64_t) * d->n_clips);
  if (e == hipSuccess) e = hipMalloc(&h->d_span, sizeof(int64_t) * d->n_clips);
  if (e == hipSuccess) e = hipMalloc(&h->d_dur, sizeof(double) * d->n_clips);
  if (e == hipSuccess) e = hipMemcpy(h->d_first, first, sizeof(int64_t) * d->n_clips, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_span, span, sizeof(int64_t) * d->n_clips, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_dur, dur, sizeof(double) * d->n_clips, hipMemcpyHostToDevice);
  delete[] first;
  delete[] span;
  delete[] dur;
  if (e != hipSuccess) {
    amp_motion_destroy(h);
    return fail(AMP_ERR_HIP, "amp_motion_create: %s", hipGetErrorString(e));
  }
  MotionView& v = h->v;
  v.clips = ClipMeta{h->d_first, h->d_span, h->d_dur, d->dt, d->n_clips};
  v.dof_pos = d->dof_positions;
  v.dof_vel = d->dof_velocities;
  v.body_pos = d->body_positions;
  v.body_rot = d->body_rotations;
  v.body_lin = d->body_linear_velocities;
  v.body_ang = d->body_angular_velocities;
  v.hot = nullptr;
  v.n_dof = d->n_dof;
  v.n_bodies = d->n_bodies;
  v.n_key = 0;
  v.D = 0;
  v.HP = 0;
  // private padded copy of the six tables for MotionLoader.sample (16-B gathers of any 4 columns).  The caller's tables may
  // have been written on any stream: one device-wide sync at create time, then the build on the null stream.
  v.SP = samp_row_floats(d->n_dof, d->n_bodies);
  e = hipMalloc(&h->d_samp, sizeof(float) * (size_t)total * v.SP);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) {
    v.samp = h->d_samp;
    const int64_t cells = total * v.SP;
    build_sample_table_kernel<<<grid_for(cells, kBlock), kBlock, 0, nullptr>>>(v, total, h->d_samp);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  }
  if (e != hipSuccess) {
    amp_motion_destroy(h);
    return fail(AMP_ERR_HIP, "amp_motion_create (sample table): %s", hipGetErrorString(e));
  }
  *out = h;
  return AMP_OK;
}

int amp_motion_destroy(AmpMotion* h) {
  if (!h) return AMP_OK;
  (void)hipFree(h->d_first);
  (void)hipFree(h->d_span);
  (void)hipFree(h->d_dur);
  (void)hipFree(h->d_hot);
  (void)hipFree(h->d_samp);
  (void)hipFree(h->d_perm);
  delete h;
  return AMP_OK;
}

int amp_motion_set_obs_layout(AmpMotion* h, const int32_t* dof_perm, int32_t ref_body, const int32_t* key_bodies,
                              int32_t n_key, amp_stream_t stream) {
  AMP_REQUIRE(h && dof_perm && key_bodies, "amp_motion_set_obs_layout: null argument");
  AMP_REQUIRE(n_key >= 1 && n_key <= kMaxKey, "amp_motion_set_obs_layout: n_key must be in [1, %d]", kMaxKey);
  AMP_REQUIRE(ref_body >= 0 && ref_body < h->v.n_bodies, "amp_motion_set_obs_layout: ref_body %d out of range", ref_body);
  for (int i = 0; i < h->v.n_dof; ++i)
    AMP_REQUIRE(dof_perm[i] >= 0 && dof_perm[i] < h->v.n_dof, "amp_motion_set_obs_layout: dof_perm[%d]=%d out of range", i,
                dof_perm[i]);
  for (int i = 0; i < n_key; ++i)
    AMP_REQUIRE(key_bodies[i] >= 0 && key_bodies[i] < h->v.n_bodies, "amp_motion_set_obs_layout: key body %d out of range",
                key_bodies[i]);
  const int D = 2 * h->v.n_dof + 13 + 3 * n_key;
  (void)hipFree(h->d_hot);
  (void)hipFree(h->d_perm);
  h->d_hot = nullptr;
  h->d_perm = nullptr;
  const int HP = (D + 3) / 4 * 4;
  AMP_HIP(hipMalloc(&h->d_hot, sizeof(float) * h->n_frames * HP));
  AMP_HIP(hipMalloc(&h->d_perm, sizeof(int32_t) * (h->v.n_dof + kMaxKey)));
  hipStream_t st = (hipStream_t)stream;
  AMP_HIP(hipMemcpyAsync(h->d_perm, dof_perm, sizeof(int32_t) * h->v.n_dof, hipMemcpyHostToDevice, st));
  AMP_HIP(hipMemcpyAsync(h->d_perm + h->v.n_dof, key_bodies, sizeof(int32_t) * n_key, hipMemcpyHostToDevice, st));
  AMP_HIP(hipStreamSynchronize(st));  // the host arrays may be temporaries of the caller
  h->ref_body = ref_body;
  for (int i = 0; i < n_key; ++i) h->key_bodies[i] = key_bodies[i];
  h->v.n_key = n_key;
  h->v.D = D;
  h->v.HP = HP;
  h->v.hot = h->d_hot;
  const int64_t total = h->n_frames * h->v.HP;
  { amp::TraceScope trace__("build_hot_kernel", st);
    build_hot_kernel<<<grid_for(total, kBlock), kBlock, 0, st>>>(h->v, h->n_frames, h->d_perm, ref_body, h->d_perm + h->v.n_dof,
                                                              h->d_hot);
  }
  int rc = launch_status("build_hot_kernel");
  if (rc != AMP_OK) return rc;
  h->has_layout = true;
  return AMP_OK;
}

int amp_motion_frame_blend(const AmpMotion* h, const double* times, const int64_t* ids, int64_t n, int64_t* i0, int64_t* i1,
                           double* blend, amp_stream_t stream) {
  AMP_REQUIRE(h, "amp_motion_frame_blend: null handle");
  AMP_REQUIRE(n >= 0, "amp_motion_frame_blend: negative n");
  if (n == 0) return AMP_OK;
  AMP_REQUIRE(times && i0 && i1 && blend, "amp_motion_frame_blend: null buffer");
  { amp::TraceScope trace__("frame_blend_kernel", (hipStream_t)stream);
    frame_blend_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(h->v.clips, times, ids, n, i0, i1, blend);
  }
  return launch_status("frame_blend_kernel");
}

int amp_motion_sample(const AmpMotion* h, const double* times, const int64_t* ids, int64_t n, float* dp, float* dv, float* bp,
                      float* br, float* bl, float* ba, amp_stream_t stream) {
  AMP_REQUIRE(h, "amp_motion_sample: null handle");
  AMP_REQUIRE(n >= 0, "amp_motion_sample: negative n");
  if (n == 0) return AMP_OK;
  AMP_REQUIRE(times, "amp_motion_sample: times is null");
  AMP_REQUIRE(n * (int64_t)h->v.n_bodies * 4 < ((int64_t)1 << 40), "amp_motion_sample: n too large");
  const int T = sample_tile(h->v.n_dof, h->v.n_bodies);
  const size_t lds = sample_lds(T, h->v.n_dof, h->v.n_bodies);
  AMP_REQUIRE(lds <= 64 * 1024, "amp_motion_sample: %d bodies x %d DoFs need %zu B of LDS per 8-sample tile (> 64 KiB)", h->v.n_bodies,
              h->v.n_dof, lds);
  { amp::TraceScope trace__("sample_kernel", (hipStream_t)stream);
    sample_kernel<<<grid_for(n, T), kBlock, lds, (hipStream_t)stream>>>(h->v, times, ids, n, T, SampleOuts{{dp, dv, bp, bl, ba, br}});
  }
  return launch_status("sample_kernel");
}

int amp_collect_reference(const AmpMotion* h, const double* times, const int64_t* ids, int64_t n, int32_t K, float* out,
                          const int64_t* dst_rows, amp_stream_t stream) {
  AMP_REQUIRE(h, "amp_collect_reference: null handle");
  AMP_REQUIRE(h->has_layout, "amp_collect_reference: call amp_motion_set_obs_layout first");
  AMP_REQUIRE(n >= 0 && K >= 1, "amp_collect_reference: need n >= 0 and K >= 1");
  if (n == 0) return AMP_OK;
  AMP_REQUIRE(times && out, "amp_collect_reference: null buffer");
  { amp::TraceScope trace__("collect_reference_kernel", (hipStream_t)stream);
    if (!dst_rows && n * K < (int64_t)1 << 31 && expert_wide_lds(h->v.D) <= 64 * 1024)
      collect_reference_wide_kernel<<<grid_for(n * K, kExpertWide), kBlock, expert_wide_lds(h->v.D), (hipStream_t)stream>>>(h->v, times, ids, n, K, out);
    else
      collect_reference_kernel<<<grid_for(n * K, kExpertTile), kBlock, expert_lds(h->v.D), (hipStream_t)stream>>>(h->v, times, ids, n, K, out, dst_rows, nullptr);
  }
  return launch_status("collect_reference_kernel");
}

int amp_motion_sample_times(const AmpMotion* h, uint64_t seed, uint64_t step, int32_t start, const int64_t* index,
                            const int64_t* n_dev, int64_t n, int64_t* motion_ids, double* times, amp_stream_t stream) {
  AMP_REQUIRE(h, "amp_motion_sample_times: null handle");
  AMP_REQUIRE(n >= 0, "amp_motion_sample_times: negative n");
  if (n == 0) return AMP_OK;
  AMP_REQUIRE(motion_ids && times, "amp_motion_sample_times: null buffer");
  { amp::TraceScope trace__("sample_times_kernel", (hipStream_t)stream);
    sample_times_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(h->v.clips, seed, step, start, index, n, n_dev,
                                                                              motion_ids, times, nullptr, nullptr, 0, ResetClears{}, nullptr);
  }
  return launch_status("sample_times_kernel");
}

int amp_reset_apply(const AmpMotion* h, const AmpResetArgs* a, amp_stream_t stream) {
  AMP_REQUIRE(h && a, "amp_reset_apply: null argument");
  AMP_REQUIRE(h->has_layout, "amp_reset_apply: call amp_motion_set_obs_layout first");
  AMP_REQUIRE(a->max_n >= 0 && a->K >= 1, "amp_reset_apply: need max_n >= 0 and K >= 1");
  if (a->max_n == 0) return AMP_OK;
  AMP_REQUIRE(a->mode == AMP_RESET_REFERENCE, "amp_reset_apply: only the reference-motion reset (mode 0); the default strategy is served by amp_reset_compact_apply");
  AMP_REQUIRE(!a->step_dev_out, "amp_reset_apply: step_dev_out is served by amp_reset_compact_apply only");
  AMP_REQUIRE(a->env_ids && a->count && a->motion_ids && a->motion_times, "amp_reset_apply: null buffer");
  AMP_REQUIRE(!a->last_actions || a->n_actions >= 1, "amp_reset_apply: last_actions needs n_actions >= 1");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = a->max_n;
  { amp::TraceScope trace__("sample_times_kernel", st);
    sample_times_kernel<<<grid_for(n, kBlock), kBlock, 0, st>>>(h->v.clips, a->seed, a->step, a->start, a->env_ids, n, a->count,
                                                             a->motion_ids, a->motion_times, a->env_motion_ids,
                                                             a->env_motion_start_times, a->env_offset,
                                                             ResetClears{a->episode_length, a->last_actions, a->just_reset, a->n_actions},
                                                             a->step_dev);
  }
  int rc = launch_status("sample_times_kernel");
  if (rc != AMP_OK) return rc;
  if (a->root_state || a->dof_pos || a->dof_vel) {
    amp::TraceScope trace__("reset_state_kernel", st);
    reset_state_kernel<<<grid_for(n, kBlock), kBlock, 0, st>>>(h->v, a->motion_times, a->motion_ids, a->env_ids, n, a->env_origins,
                                                            a->z_lift, a->root_state, a->dof_pos, a->dof_vel, a->count);
  }
  rc = launch_status("reset_state_kernel");
  if (rc != AMP_OK) return rc;
  if (a->amp_obs_buffer) {
    amp::TraceScope trace__("collect_reference_kernel", st);
    collect_reference_kernel<<<grid_for(n * a->K, kExpertTile), kBlock, expert_lds(h->v.D), st>>>(h->v, a->motion_times, a->motion_ids, n, a->K,
                                                                              a->amp_obs_buffer, a->env_ids, a->count);
  }
  return launch_status("collect_reference_kernel");
}

int amp_reset_compact_apply(const AmpMotion* h, const AmpCompactArgs* c, const AmpResetArgs* a, const AmpCommandArgs* cmd,
                            const AmpRewardLogArgs* lg, amp_stream_t stream) {
  AMP_REQUIRE(h && c && a, "amp_reset_compact_apply: null argument");
  AMP_REQUIRE(h->has_layout, "amp_reset_compact_apply: call amp_motion_set_obs_layout first");
  AMP_REQUIRE(c->num_envs >= 0 && a->K >= 1, "amp_reset_compact_apply: need num_envs >= 0 and K >= 1");
  AMP_REQUIRE(c->count, "amp_reset_compact_apply: count pointer is null");
  const int64_t N = c->num_envs;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    AMP_HIP(hipMemsetAsync(c->count, 0, sizeof(int64_t), st));
    return AMP_OK;
  }
  AMP_REQUIRE(c->mask && c->tile_counts && c->ids, "amp_reset_compact_apply: null compaction buffer");
  AMP_REQUIRE(c->tile_envs == 8 || c->tile_envs == 16 || c->tile_envs == 32 || c->tile_envs == 64,
              "amp_reset_compact_apply: tile_envs must be 8, 16, 32 or 64");
  AMP_REQUIRE(a->env_ids == c->ids && a->count == c->count && a->max_n >= N,
              "amp_reset_compact_apply: the reset arguments must consume the compaction's ids / count (max_n >= num_envs)");
  AMP_REQUIRE(a->mode == AMP_RESET_REFERENCE || a->mode == AMP_RESET_DEFAULT, "amp_reset_compact_apply: unknown reset mode %d", a->mode);
  AMP_REQUIRE(!a->step_dev_out || (a->step_dev && a->step_dev_out != a->step_dev),
              "amp_reset_compact_apply: step_dev_out needs step_dev, and a different word");
  if (a->mode == AMP_RESET_DEFAULT) {
    AMP_REQUIRE(a->default_root_state && a->default_joint_pos && a->default_joint_vel,
                "amp_reset_compact_apply: AMP_RESET_DEFAULT needs default_root_state / default_joint_pos / default_joint_vel");
    AMP_REQUIRE(!cmd, "amp_reset_compact_apply: the default reset strategy does not resample commands (g1_amp_env.py:362-369)");
  } else {
    AMP_REQUIRE(a->motion_ids && a->motion_times && a->amp_obs_buffer, "amp_reset_compact_apply: null buffer (motion_ids / motion_times / amp_obs_buffer)");
  }
  AMP_REQUIRE(!lg || (lg->reward_terms && lg->means && lg->n_terms >= 1 && lg->n_terms <= 64),
              "amp_reset_compact_apply: the reward-log arguments need reward_terms, means and 1..64 terms");
  AMP_REQUIRE(!a->last_actions || a->n_actions >= 1, "amp_reset_compact_apply: last_actions needs n_actions >= 1");
  AMP_REQUIRE(!cmd || (cmd->command && cmd->time_left), "amp_reset_compact_apply: null command buffer");
  AMP_REQUIRE(!cmd || !(cmd->vel_span > 0.0f) || cmd->t_span >= 0.0f, "amp_reset_compact_apply: negative resampling-time span");
  const int sub = kTile / c->tile_envs;
  const int64_t n_counts = (N + c->tile_envs - 1) / c->tile_envs;
  const int64_t n_tiles = (N + kTile - 1) / kTile;
  const unsigned grid = (unsigned)((n_tiles + 3) / 4);
  const size_t lds = expert_lds(h->v.D);
  AMP_REQUIRE(lds <= 48 * 1024, "amp_reset_compact_apply: expert tile needs %zu B of LDS", lds);
  { amp::TraceScope trace__("reset_compact_apply_kernel", st);
    reset_compact_apply_kernel<<<grid + (lg ? (unsigned)lg->n_terms : 0u), kBlock, lds, st>>>(h->v, *c, *a, cmd ? *cmd : AmpCommandArgs{}, cmd ? 1 : 0,
                                                                                           n_tiles, sub, n_counts, grid, lg ? *lg : AmpRewardLogArgs{});
  }
  return launch_status("reset_compact_apply_kernel");
}

int amp_reset_reference_state(const AmpMotion* h, const double* times, const int64_t* ids, const int64_t* env_ids, int64_t n,
                              const float* origins, float z_lift, float* root, float* dp, float* dv, amp_stream_t stream) {
  AMP_REQUIRE(h, "amp_reset_reference_state: null handle");
  AMP_REQUIRE(h->has_layout, "amp_reset_reference_state: call amp_motion_set_obs_layout first");
  AMP_REQUIRE(n >= 0, "amp_reset_reference_state: negative n");
  if (n == 0) return AMP_OK;
  AMP_REQUIRE(times, "amp_reset_reference_state: times is null");
  { amp::TraceScope trace__("reset_state_kernel", (hipStream_t)stream);
    reset_state_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(h->v, times, ids, env_ids, n, origins, z_lift,
                                                                             root, dp, dv, nullptr);
  }
  return launch_status("reset_state_kernel");
}

}  // extern "C"
