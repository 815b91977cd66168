// Device-side pieces of the motion sampler that more than one translation unit launches: table views, the fp64
// frame / blend index, and the expert-observation tile (collect_reference_motions fused with compute_obs).
#pragma once
#include "amp_common.hpp"

typedef float f4 __attribute__((ext_vector_type(4)));

namespace amp {

struct ClipMeta {
  const int64_t* first;  // [C] global index of the clip's first frame
  const int64_t* span;   // [C] frames - 1
  const double* dur;     // [C] dt * (frames - 1)
  double dt;
  int32_t n_clips;
};

struct MotionView {
  ClipMeta clips;
  const float* dof_pos;
  const float* dof_vel;
  const float* body_pos;
  const float* body_rot;
  const float* body_lin;
  const float* body_ang;
  const float* hot;  // [F, D] private hot-subset table
  int32_t n_dof, n_bodies, n_key, D;
  int32_t HP;  // floats per hot row: D rounded up to a multiple of 4 (16-B aligned rows for dwordx4 gathers)
  // private copy of the six tables for MotionLoader.sample: per frame [dof_pos | dof_vel | body_pos | body_lin | body_ang |
  // body_rot], every segment padded to a multiple of 4 floats, so that any 4 columns of any table are ONE 16-B gather
  const float* samp;
  int32_t SP;  // floats per row of `samp`
};

// segment widths (floats, padded to 4) and the row pitch of the sample table
__host__ __device__ inline int samp_dof_w(int n_dof) { return (n_dof + 3) & ~3; }
__host__ __device__ inline int samp_body_w(int n_bodies) { return (3 * n_bodies + 3) & ~3; }
__host__ __device__ inline int samp_row_floats(int n_dof, int n_bodies) {
  return 2 * samp_dof_w(n_dof) + 3 * samp_body_w(n_bodies) + 4 * n_bodies;
}

}  // namespace amp

struct AmpMotion {
  amp::MotionView v;
  int device;
  int64_t n_frames;
  int64_t* d_first;
  int64_t* d_span;
  double* d_dur;
  float* d_hot;
  float* d_samp;
  int32_t* d_perm;
  int32_t ref_body;
  int32_t key_bodies[amp::kMaxKey];
  bool has_layout;
};

namespace amp {

// motions/motion_loader.py:281-307 in fp64 with numpy semantics.
__device__ __forceinline__ void frame_blend_ref(const ClipMeta& m, double t, int64_t clip, int64_t& i0, int64_t& i1,
                                                double& blend) {
  // out-of-range ids would index the clip arrays out of bounds: clamp (the host wrapper validates too)
  clip = clip < 0 ? 0 : (clip >= m.n_clips ? m.n_clips - 1 : clip);
  const double dur = m.dur[clip];
  const int64_t first = m.first[clip];
  const int64_t span = m.span[clip];
  double phase = t / dur;
  phase = phase < 0.0 ? 0.0 : (phase > 1.0 ? 1.0 : phase);  // np.clip
  if (!(phase >= 0.0)) phase = 0.0;                           // NaN time: keep the index in range
  const int64_t l0 = (int64_t)rint(phase * (double)span);    // round-half-to-even
  const int64_t l1 = l0 + 1 < span ? l0 + 1 : span;
  i0 = first + l0;
  i1 = first + l1;
  blend = rint(((t - (double)l0 * m.dt) / m.dt) * 1e5) / 1e5;  // == np.round(x, 5)
}

// ------------------------------------------------------------------------------------------------
// collect_reference_motions fused with compute_obs (g1_amp_env.py:445-486, 535-561)
// ------------------------------------------------------------------------------------------------
struct ExpertSlot {
  int32_t i0, i1;
  float blend;
  float tn[6];   // tangent | normal of the slerped reference-body quaternion
  float rp[3];   // lerped reference-body position
  int64_t obase; // float offset of this sample's output row
};

constexpr int kExpertTile = 64;  // samples per workgroup: many short workgroups hide the L2 gather latency

// Optional by-products of the newest (k = 0) frame of every row, for the one-launch device reset (motion.hip): the
// reference root state (g1_amp_env.py:385-411) falls out of phase A's gathers, the robot-order DoF rows are the first
// 2 n_dof columns of the k = 0 expert frame (the same lerp_ref of the same hot columns: bit-identical to reset_state_kernel).
struct ResetRows {
  float* root_state;        // [*, 13] compact rows (slot of row r = slots[r]), may be NULL
  float* dof_pos;           // [*, n_dof], may be NULL
  float* dof_vel;
  const int64_t* slots;     // [n] compact slot of row r
  const float* origins;     // [num_envs, 3] indexed by dst_rows[r], or NULL
  float z_lift;
};

// Phase A: one sample per lane (fp64 frame/blend, SLERP of the reference quaternion, tangent/normal).
// Phase B: a lane LERPs FOUR consecutive columns of one sample from two 16-B gathers (hot rows are padded to a
//          16-B pitch) and drops them into an LDS image of the tile's output, which is contiguous in HBM
//          ([64 samples][D floats], 16-B aligned) unless rows are scattered (dst_rows).
// Phase C: the image is streamed out with 16-B non-temporal stores, 1 KiB per wave instruction (the expert rows feed the
//          discriminator update, not the next kernels of the step).
// `block` = index of the 64-sample tile, `s_img` = the workgroup's dynamic LDS (16-B aligned, expert_lds(D) bytes):
// a device function so that it can also run as part of a horizontally fused launch (env_step.hip).
template <bool kReset = false>
__device__ __forceinline__ void collect_reference_body(const MotionView& v, const double* __restrict__ times,
                                                       const int64_t* __restrict__ ids, int64_t n, int K,
                                                       float* __restrict__ out, const int64_t* __restrict__ dst_rows,
                                                       const int64_t* __restrict__ n_dev, int64_t block, float* s_img,
                                                       const ResetRows* rr = nullptr) {
  // s_img: [64][D] output image, then the slots
  const int D = v.D, HP = v.HP, nd2 = 2 * v.n_dof;
  ExpertSlot* slots = reinterpret_cast<ExpertSlot*>(s_img + ((kExpertTile * D + 3) & ~3));
  if (n_dev) n = *n_dev < n ? *n_dev : n;  // device-side sample count (reset path without a host read-back)
  const int64_t total = n * K;
  const int64_t tile_base = block * kExpertTile;
  if (tile_base >= total) return;  // uniform for the whole workgroup
  const int n_tile = (int)((total - tile_base) < kExpertTile ? (total - tile_base) : kExpertTile);
  const float* __restrict__ hot = v.hot;
  if (threadIdx.x < n_tile) {
    const int64_t sidx = tile_base + threadIdx.x;
    const int64_t r = sidx / K;
    const int k = (int)(sidx - r * K);
    // history time t - dt*k in fp64 (g1_amp_env.py:454-457)
    const double t = times[r] - v.clips.dt * (double)k;
    int64_t a, b;
    double w;
    frame_blend_ref(v.clips, t, ids ? ids[r] : 0, a, b, w);
    ExpertSlot sl;
    sl.i0 = (int32_t)a;
    sl.i1 = (int32_t)b;
    sl.blend = (float)w;
    const float* r0 = hot + a * HP + nd2;
    const float* r1 = hot + b * HP + nd2;
    sl.rp[0] = lerp_ref(r0[0], r1[0], sl.blend);
    sl.rp[1] = lerp_ref(r0[1], r1[1], sl.blend);
    sl.rp[2] = lerp_ref(r0[2], r1[2], sl.blend);
    const Quat q = slerp_ref(Quat{r0[3], r0[4], r0[5], r0[6]}, Quat{r1[3], r1[4], r1[5], r1[6]}, sl.blend);
    const Vec3 tg = quat_apply_ref(q, Vec3{1.0f, 0.0f, 0.0f});
    const Vec3 nm = quat_apply_ref(q, Vec3{0.0f, 0.0f, 1.0f});
    sl.tn[0] = tg.x; sl.tn[1] = tg.y; sl.tn[2] = tg.z;
    sl.tn[3] = nm.x; sl.tn[4] = nm.y; sl.tn[5] = nm.z;
    const int64_t row = dst_rows ? dst_rows[r] : r;
    sl.obase = (row * K + k) * (int64_t)D;
    slots[threadIdx.x] = sl;
    if (kReset && k == 0 && rr->root_state) {  // reset_state_kernel's root row
      float* o = rr->root_state + rr->slots[r] * 13;
      const float* og = rr->origins ? rr->origins + row * 3 : nullptr;
      o[0] = og ? sl.rp[0] + og[0] : sl.rp[0];
      o[1] = og ? sl.rp[1] + og[1] : sl.rp[1];
      o[2] = (og ? sl.rp[2] + og[2] : sl.rp[2]) + rr->z_lift;
      o[3] = q.w; o[4] = q.x; o[5] = q.y; o[6] = q.z;
      for (int c = 7; c < 13; ++c) o[c] = lerp_ref(r0[c], r1[c], sl.blend);
    }
  }
  __syncthreads();
  // ---- phase B: (sample, quad) items; both gathers of an item are 16-B loads --------------------------
  const int QP = HP >> 2;  // quads per row
  const f4* __restrict__ hot4 = reinterpret_cast<const f4*>(hot);
  for (int it = threadIdx.x; it < n_tile * QP; it += kBlock) {
    const int s = it / QP, q = it - s * QP;
    const ExpertSlot& sl = slots[s];
    const f4 a = hot4[(int64_t)sl.i0 * QP + q];
    const f4 b = hot4[(int64_t)sl.i1 * QP + q];
    float* img = s_img + s * D;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = 4 * q + c;
      if (j >= D) break;
      float val;
      if (j == nd2) {
        val = sl.rp[2];                       // root height = lerped z (bit-identical to lerp of the z column)
      } else if (j <= nd2 + 6 && j > nd2) {
        val = sl.tn[j - nd2 - 1];             // tangent | normal
      } else {
        val = lerp_ref(a[c], b[c], sl.blend);
        if (j >= nd2 + 13) val = val - sl.rp[(j - nd2 - 13) % 3];  // key body relative to the reference body (:552)
      }
      img[j] = val;
    }
  }
  __syncthreads();
  // ---- phase C: stream the image out -------------------------------------------------------------------
  if (!dst_rows) {
    // rows of consecutive samples are contiguous: the whole tile is one 16-B aligned run (64 * D floats)
    float* dst = out + tile_base * D;
    const int count = n_tile * D;
    if ((count & 3) == 0 && (((uintptr_t)dst) & 15) == 0) {
      const f4* src4 = reinterpret_cast<const f4*>(s_img);
      f4* dst4 = reinterpret_cast<f4*>(dst);
      for (int e = threadIdx.x; e < (count >> 2); e += kBlock) __builtin_nontemporal_store(src4[e], dst4 + e);
    } else {
      for (int e = threadIdx.x; e < count; e += kBlock) dst[e] = s_img[e];
    }
  } else {
    // scattered rows (reset path): every K-run of samples of one env is contiguous; store per sample
    for (int e = threadIdx.x; e < n_tile * D; e += kBlock) {
      const int s = e / D, j = e - s * D;
      out[slots[s].obase + j] = s_img[e];
    }
  }
  if (kReset && (rr->dof_pos || rr->dof_vel)) {
    // robot-order DoF rows of the reset state = columns [0, nd) / [nd, 2 nd) of the k = 0 frames of this tile
    const int nd = v.n_dof;
    for (int e = threadIdx.x; e < n_tile * nd2; e += kBlock) {
      const int s = e / nd2, j = e - s * nd2;
      const int64_t sidx = tile_base + s;
      const int64_t r = sidx / K;
      if (sidx - r * K != 0) continue;
      float* dst = j < nd ? rr->dof_pos : rr->dof_vel;
      if (dst) dst[rr->slots[r] * nd + (j < nd ? j : j - nd)] = s_img[s * D + j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Wide tile body of the same sample (contiguous output rows, fewer than 2^31 samples: the hot path).  The body above
// spends a 64-sample workgroup's life in phase A's dependent chain (times -> clip table -> two row gathers -> SLERP)
// on one wave, and ~90 VALU instructions per 16 B of phase B on row/column arithmetic and per-column branches.  Here
//   * a workgroup owns 256 samples: phase A runs once, one sample per lane on all four waves;
//   * the 256 samples go through the LDS image in four 64-sample passes.  In phase B a lane owns ONE quad of columns
//     (its gather offset, which of its columns are key-body columns, where they land) and walks the pass's samples:
//     the key-body subtraction reads a per-sample strip {0,0,0,0, rp, rp} at a per-lane offset (non-key columns
//     subtract +0.0f, which changes no bit), and the seven SLERP-derived columns are redirected to a scratch word
//     and filled from phase A's values by a separate 448-element pass -- no per-column branch in the loop.
// Bit-identical to collect_reference_body (tests/test_gpu_motion.py).
// ------------------------------------------------------------------------------------------------
constexpr int kExpertWide = 256;
constexpr int kExpertStrip = 11;  // floats per sample of the subtraction strip (10 used; odd pitch)

__host__ __device__ inline int expert_wide_lds_floats(int D) {
  return ((kExpertTile * D + 3) & ~3) + kExpertWide * 4 + kExpertWide * kExpertStrip + kExpertWide * 7 + 4;
}

__device__ __forceinline__ void collect_reference_wide_body(const MotionView& v, const double* __restrict__ times,
                                                            const int64_t* __restrict__ ids, int64_t n, int K,
                                                            float* __restrict__ out, int64_t block, float* smem) {
  const int D = v.D, HP = v.HP, nd2 = 2 * v.n_dof, QP = HP >> 2;
  const int tid = threadIdx.x;
  float* s_img = smem;                                   // [64][D] output image of one pass
  float* s_slot = s_img + ((kExpertTile * D + 3) & ~3);  // [256][4] i0 | i1 | blend | -
  float* s_strip = s_slot + kExpertWide * 4;             // [256][11] 0 0 0 0 | rp | rp
  float* s_tn = s_strip + kExpertWide * kExpertStrip;    // [256][7] height | tangent | normal
  float* s_trash = s_tn + kExpertWide * 7;               // [4]
  const int64_t total = n * K;
  const int64_t tile_base = block * kExpertWide;
  if (tile_base >= total) return;  // uniform for the whole workgroup
  const int n_tile = (int)((total - tile_base) < kExpertWide ? (total - tile_base) : kExpertWide);
  const float* __restrict__ hot = v.hot;
  // ---- phase A: one sample per lane (fp64 frame/blend, SLERP of the reference quaternion, tangent/normal) -------
  if (tid < n_tile) {
    const uint32_t sidx = (uint32_t)(tile_base + tid);  // total < 2^31 (host)
    const uint32_t r = sidx / (uint32_t)K;
    const int k = (int)(sidx - r * (uint32_t)K);
    const double t = times[r] - v.clips.dt * (double)k;  // history time t - dt*k in fp64 (g1_amp_env.py:454-457)
    int64_t a, b;
    double w;
    frame_blend_ref(v.clips, t, ids ? ids[r] : 0, a, b, w);
    const float blend = (float)w;
    const float* r0 = hot + a * HP + nd2;
    const float* r1 = hot + b * HP + nd2;
    const float p0 = lerp_ref(r0[0], r1[0], blend), p1 = lerp_ref(r0[1], r1[1], blend), p2 = lerp_ref(r0[2], r1[2], blend);
    const Quat q = slerp_ref(Quat{r0[3], r0[4], r0[5], r0[6]}, Quat{r1[3], r1[4], r1[5], r1[6]}, blend);
    const Vec3 tg = quat_apply_ref(q, Vec3{1.0f, 0.0f, 0.0f});
    const Vec3 nm = quat_apply_ref(q, Vec3{0.0f, 0.0f, 1.0f});
    f4 sl;
    sl[0] = __int_as_float((int32_t)a); sl[1] = __int_as_float((int32_t)b); sl[2] = blend; sl[3] = 0.0f;
    *reinterpret_cast<f4*>(s_slot + tid * 4) = sl;
    float* st = s_strip + tid * kExpertStrip;
    st[0] = 0.0f; st[1] = 0.0f; st[2] = 0.0f; st[3] = 0.0f;
    st[4] = p0; st[5] = p1; st[6] = p2; st[7] = p0; st[8] = p1; st[9] = p2;
    float* tn = s_tn + tid * 7;
    tn[0] = p2;  // root height = lerped z
    tn[1] = tg.x; tn[2] = tg.y; tn[3] = tg.z; tn[4] = nm.x; tn[5] = nm.y; tn[6] = nm.z;
  }
  __syncthreads();
  // ---- per-lane constants of phase B: the quad this lane owns --------------------------------------------
  const int G = kBlock / QP, g = (int)(((float)tid + 0.5f) * (1.0f / (float)QP)), q = tid - g * QP;
  const bool walker = g < G;
  const int j0 = 4 * q, jk = nd2 + 13;  // first key-body column (:552)
  const int strip0 = j0 + 3 < jk ? 0 : (j0 >= jk ? 4 + (j0 - jk) % 3 : 4 - (jk - j0));
  int woff[4], wstep[4];  // float offsets from smem: where column j0 + c lands / how far it moves per sample step
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = j0 + c;
    const bool keep = j < D && !(j >= nd2 && j <= nd2 + 6);
    woff[c] = keep ? g * D + j : (int)(s_trash - smem) + c;
    wstep[c] = keep ? G * D : 0;
  }
  const f4* __restrict__ hot4 = reinterpret_cast<const f4*>(hot);
#pragma unroll 1
  for (int u = 0; u < n_tile; u += kExpertTile) {
    const int n_sub = n_tile - u < kExpertTile ? n_tile - u : kExpertTile;
    if (walker) {
      float* w0 = smem + woff[0]; float* w1 = smem + woff[1]; float* w2 = smem + woff[2]; float* w3 = smem + woff[3];
      const float* slot = s_slot + (u + g) * 4;
      const float* strip = s_strip + (u + g) * kExpertStrip + strip0;
#pragma unroll 2
      for (int s = g; s < n_sub; s += G) {
        const f4 sl = *reinterpret_cast<const f4*>(slot);
        const f4 a = hot4[(int64_t)__float_as_int(sl[0]) * QP + q];
        const f4 b = hot4[(int64_t)__float_as_int(sl[1]) * QP + q];
        const float t = sl[2];
        // lerp_ref, then the key-body columns relative to the reference body (:552); others subtract +0.0f
        *w0 = lerp_ref(a[0], b[0], t) - strip[0];
        *w1 = lerp_ref(a[1], b[1], t) - strip[1];
        *w2 = lerp_ref(a[2], b[2], t) - strip[2];
        *w3 = lerp_ref(a[3], b[3], t) - strip[3];
        w0 += wstep[0]; w1 += wstep[1]; w2 += wstep[2]; w3 += wstep[3];
        slot += G * 4;
        strip += G * kExpertStrip;
      }
    }
    for (int e = tid; e < n_sub * 7; e += kBlock) {  // the SLERP-derived columns nd2 .. nd2 + 6
      const int s = (int)(((float)e + 0.5f) * (1.0f / 7.0f)), k = e - 7 * s;
      s_img[s * D + nd2 + k] = s_tn[(u + s) * 7 + k];
    }
    __syncthreads();
    // ---- stream the pass's image out: rows of consecutive samples are one contiguous run ------------------
    float* dst = out + (tile_base + u) * D;
    const int count = n_sub * D;
    if ((count & 3) == 0 && (((uintptr_t)dst) & 15) == 0) {
      const f4* src4 = reinterpret_cast<const f4*>(s_img);
      f4* dst4 = reinterpret_cast<f4*>(dst);
      for (int e = tid; e < (count >> 2); e += kBlock) __builtin_nontemporal_store(src4[e], dst4 + e);
    } else {
      for (int e = tid; e < count; e += kBlock) dst[e] = s_img[e];
    }
    __syncthreads();  // the image is free for the next pass
  }
}

static inline size_t expert_wide_lds(int D) { return sizeof(float) * (size_t)expert_wide_lds_floats(D); }

static inline size_t expert_lds(int D) { return sizeof(float) * (size_t)((kExpertTile * D + 3) & ~3) + sizeof(ExpertSlot) * kExpertTile; }

}  // namespace amp
