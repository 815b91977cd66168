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
};

}  // namespace amp

struct AmpMotion {
  amp::MotionView v;
  int device;
  int64_t n_frames;
  int64_t* d_first;
  int64_t* d_span;
  double* d_dur;
  float* d_hot;
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

// Phase A: one sample per lane (fp64 frame/blend, SLERP of the reference quaternion, tangent/normal).
// Phase B: a lane LERPs FOUR consecutive columns of one sample from two 16-B gathers (hot rows are padded to a
//          16-B pitch) and drops them into an LDS image of the tile's output, which is contiguous in HBM
//          ([64 samples][D floats], 16-B aligned) unless rows are scattered (dst_rows).
// Phase C: the image is streamed out with 16-B stores, 1 KiB per wave instruction.
// `block` = index of the 64-sample tile, `s_img` = the workgroup's dynamic LDS (16-B aligned, expert_lds(D) bytes):
// a device function so that it can also run as part of a horizontally fused launch (env_step.hip).
__device__ __forceinline__ void collect_reference_body(const MotionView& v, const double* __restrict__ times,
                                                       const int64_t* __restrict__ ids, int64_t n, int K,
                                                       float* __restrict__ out, const int64_t* __restrict__ dst_rows,
                                                       const int64_t* __restrict__ n_dev, int64_t block, float* s_img) {
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
      for (int e = threadIdx.x; e < (count >> 2); e += kBlock) dst4[e] = src4[e];
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
}

static inline size_t expert_lds(int D) { return sizeof(float) * (size_t)((kExpertTile * D + 3) & ~3) + sizeof(ExpertSlot) * kExpertTile; }

}  // namespace amp
