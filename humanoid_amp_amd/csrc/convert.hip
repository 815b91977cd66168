// CSV -> npz motion converter on the device: the table-producing step upstream of
// MotionLoader, restated from motions/data_convert.py of the reference (cited per kernel) as five batched kernels --
// one thread per output element or per frame -- instead of per-frame Pinocchio calls and per-(frame, body) Python loops.
//
//   upsample   30 -> 60 fps: N = 2 N0 - 1 frames on numpy's linspace grids; root position and joint angles by scipy's
//              interp1d formula (float32 difference, float64 slope and result: data_convert.py:204-232), root
//              orientation by scipy's Slerp (rotation-vector form, fp64).
//   fk         forward kinematics over the URDF's joint tree (fp64; one thread per frame), link poses -> float32
//              positions and (w, x, y, z) quaternions with Eigen's matrix -> quaternion branches (:327-356).
//   diff       central differences, one-sided at the ends (:284-287, :358-362), in the array's own precision.
//   gauss      scipy gaussian_filter1d(sigma = 1): radius 4, 'reflect' boundary, symmetric-pair summation in double,
//              result in the array's precision (:289, :365, :381).
//   angvel     quaternion-difference angular velocity of adjacent frames averaged over both neighbours (:85-109,
//              :367-380), in the float32 / float64 mix of either numpy generation (see oracle/convert.py).
//
// Parity: the reference's own shipped clips are this pipeline's outputs (motions/G1_walk.npz = rows [100:300] of
// datasets/walk1_subject1.csv, motions/custom_motion.npz = rows [110:265]): tests/test_gpu_convert.py.
#include "amp_common.hpp"

struct AmpConverter {
  int32_t n_joints, n_dof, n_bodies;
  // device copies of the kinematic model, joints in parent-before-child order
  int32_t* parent;     // [J] index of the joint whose child link is this joint's parent link, -1 = root link
  int32_t* qidx;       // [J] column of the joint angle, -1 = fixed joint
  double* origin_rot;  // [J, 9] row-major rotation of the joint origin (URDF rpy)
  double* origin_xyz;  // [J, 3]
  double* axis;        // [J, 3]
  int32_t* body_joint; // [B] joint whose child link is the body, -1 = root link
};

namespace amp {

constexpr int kMaxJoints = 64;
constexpr int kGaussRadius = 4;

__device__ __forceinline__ double grid_time(int64_t i, int64_t n, double step, double stop) {
  return i == n - 1 ? stop : (double)i * step;  // numpy.linspace: arange * step, last sample = stop exactly
}

// ---- upsample ---------------------------------------------------------------------------------------------------
// csv [N0, C] float32: cols 0-2 root xyz, 3-6 root quaternion (x, y, z, w), 7.. joint angles.
// One thread per (new frame, column c != 3..6) for the linear part; threads with c == 3 do the Slerp of that frame.
__global__ __launch_bounds__(kBlock) void convert_upsample_kernel(const float* __restrict__ csv, int64_t n0, int cols,
                                                                  double* __restrict__ root_pos, double* __restrict__ root_quat,
                                                                  double* __restrict__ joints) {
  const int64_t n = 2 * n0 - 1;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n * cols) return;
  const int64_t i = e / cols;
  const int c = (int)(e - i * cols);
  if (c > 3 && c < 7) return;
  const double stop = (double)(n0 - 1) * (1.0 / 30.0);  // (N_orig - 1) * dt_orig, dt_orig = 1.0 / 30 (:197-199)
  const double step0 = n0 > 1 ? stop / (double)(n0 - 1) : 0.0, step1 = n > 1 ? stop / (double)(n - 1) : 0.0;
  const double x = grid_time(i, n, step1, stop);
  // searchsorted(t0, x, 'left') clipped to [1, n0 - 1]: t1[2 j] == t0[j] exactly, so the index is ceil(i / 2)
  int64_t hi = (i + 1) / 2;
  hi = hi < 1 ? 1 : (hi > n0 - 1 ? n0 - 1 : hi);
  const int64_t lo = hi - 1;
  const double x_lo = grid_time(lo, n0, step0, stop), x_hi = grid_time(hi, n0, step0, stop);
  if (c != 3) {
    // scipy interp1d._call_linear: slope = (y_hi - y_lo) / (x_hi - x_lo) with the float32 difference; y = slope (x - x_lo) + y_lo
    const float y_lo = csv[lo * cols + c], y_hi = csv[hi * cols + c];
    const double slope = (double)(y_hi - y_lo) / (x_hi - x_lo);
    const double y = slope * (x - x_lo) + (double)y_lo;
    if (c < 3) root_pos[i * 3 + c] = y;
    else joints[i * (cols - 7) + (c - 7)] = y;
    return;
  }
  // scipy Slerp: ind = searchsorted(times, x, 'right') - 1 (0 at the first sample, n0 - 2 at the last),
  // alpha = (x - t[ind]) / (t[ind + 1] - t[ind]); result = R[ind] * exp(alpha * log(R[ind]^-1 R[ind + 1]))
  // (scipy: searchsorted(..., 'left') - 1, forced to 0 at the first sample: an original frame j >= 1 is reached as
  // R[j - 1] * exp(1.0 * log(...)), not copied)
  const int64_t ind = i == 0 ? 0 : (i + 1) / 2 - 1;
  const double t_a = grid_time(ind, n0, step0, stop), t_b = grid_time(ind + 1, n0, step0, stop);
  const double alpha = (x - t_a) / (t_b - t_a);
  double qa[4], qb[4];
  {
    double na = 0.0, nb = 0.0;
    for (int k = 0; k < 4; ++k) {
      qa[k] = (double)csv[ind * cols + 3 + k];
      qb[k] = (double)csv[(ind + 1) * cols + 3 + k];
      na += qa[k] * qa[k];
      nb += qb[k] * qb[k];
    }
    na = sqrt(na); nb = sqrt(nb);
    for (int k = 0; k < 4; ++k) { qa[k] /= na; qb[k] /= nb; }  // Rotation.from_quat normalises
  }
  // d = conj(qa) * qb  (x, y, z, w)
  double d[4];
  d[0] = qa[3] * qb[0] - qa[0] * qb[3] - qa[1] * qb[2] + qa[2] * qb[1];
  d[1] = qa[3] * qb[1] - qa[1] * qb[3] - qa[2] * qb[0] + qa[0] * qb[2];
  d[2] = qa[3] * qb[2] - qa[2] * qb[3] - qa[0] * qb[1] + qa[1] * qb[0];
  d[3] = qa[3] * qb[3] + qa[0] * qb[0] + qa[1] * qb[1] + qa[2] * qb[2];
  {
    const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
    for (int k = 0; k < 4; ++k) d[k] /= nd;
  }
  if (d[3] < 0.0) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; d[3] = -d[3]; }  // as_rotvec works on w >= 0
  const double sv = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const double angle = 2.0 * atan2(sv, d[3]);
  double scale;  // rotvec = scale * xyz
  if (angle <= 1e-3) {
    const double a2 = angle * angle;
    scale = 2.0 + a2 / 12.0 + 7.0 * a2 * a2 / 2880.0;
  } else {
    scale = angle / sin(angle / 2.0);
  }
  double rv[3] = {scale * d[0] * alpha, scale * d[1] * alpha, scale * d[2] * alpha};
  // from_rotvec
  const double ang = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  double s2;
  if (ang <= 1e-3) {
    const double a2 = ang * ang;
    s2 = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
  } else {
    s2 = sin(ang / 2.0) / ang;
  }
  const double p[4] = {s2 * rv[0], s2 * rv[1], s2 * rv[2], cos(ang / 2.0)};
  // r = qa * p, normalised
  double r[4];
  r[0] = qa[3] * p[0] + qa[0] * p[3] + qa[1] * p[2] - qa[2] * p[1];
  r[1] = qa[3] * p[1] + qa[1] * p[3] + qa[2] * p[0] - qa[0] * p[2];
  r[2] = qa[3] * p[2] + qa[2] * p[3] + qa[0] * p[1] - qa[1] * p[0];
  r[3] = qa[3] * p[3] - qa[0] * p[0] - qa[1] * p[1] - qa[2] * p[2];
  const double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  for (int k = 0; k < 4; ++k) root_quat[i * 4 + k] = r[k] / nr;
}

// ---- forward kinematics -------------------------------------------------------------------------------------------
struct Pose { double R[9]; double p[3]; };

__device__ __forceinline__ void mat_mul(const double* a, const double* b, double* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) o[r * 3 + c] = (a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c]) + a[r * 3 + 2] * b[6 + c];
}

// rotation matrix -> (w, x, y, z), Eigen's branches (the sign convention pin.Quaternion(R) leaves in the files)
__device__ __forceinline__ void matrix_to_quat(const double* m, double* q) {
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (m[7] - m[5]) * t;
    q[2] = (m[2] - m[6]) * t;
    q[3] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
    q[1 + i] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[k * 3 + j] - m[j * 3 + k]) * t;
    q[1 + j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    q[1 + k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
}

__global__ __launch_bounds__(64) void convert_fk_kernel(AmpConverter m, int64_t n, const double* __restrict__ root_pos,
                                                        const double* __restrict__ root_quat, const double* __restrict__ joints,
                                                        float* __restrict__ body_pos, float* __restrict__ body_rot) {
  const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (f >= n) return;
  Pose link[kMaxJoints];  // scratch: an offline tool, a few thousand frames
  Pose root;
  {
    // scipy Rotation.as_matrix of the unit quaternion (x, y, z, w)
    const double x = root_quat[f * 4], y = root_quat[f * 4 + 1], z = root_quat[f * 4 + 2], w = root_quat[f * 4 + 3];
    const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w, xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    root.R[0] = x2 - y2 - z2 + w2; root.R[1] = 2.0 * (xy - zw);      root.R[2] = 2.0 * (xz + yw);
    root.R[3] = 2.0 * (xy + zw);   root.R[4] = -x2 + y2 - z2 + w2;   root.R[5] = 2.0 * (yz - xw);
    root.R[6] = 2.0 * (xz - yw);   root.R[7] = 2.0 * (yz + xw);      root.R[8] = -x2 - y2 + z2 + w2;
    root.p[0] = root_pos[f * 3]; root.p[1] = root_pos[f * 3 + 1]; root.p[2] = root_pos[f * 3 + 2];
  }
  for (int j = 0; j < m.n_joints; ++j) {
    const Pose& par = m.parent[j] < 0 ? root : link[m.parent[j]];
    double Rj[9];
    const double* Ro = m.origin_rot + j * 9;
    if (m.qidx[j] >= 0) {
      // Rodrigues rotation about the joint axis
      const double a = joints[f * m.n_dof + m.qidx[j]];
      const double ax = m.axis[j * 3], ay = m.axis[j * 3 + 1], az = m.axis[j * 3 + 2];
      const double c = cos(a), s = sin(a), t = 1.0 - c;
      const double Ra[9] = {t * ax * ax + c,      t * ax * ay - s * az, t * ax * az + s * ay,
                            t * ax * ay + s * az, t * ay * ay + c,      t * ay * az - s * ax,
                            t * ax * az - s * ay, t * ay * az + s * ax, t * az * az + c};
      mat_mul(Ro, Ra, Rj);
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) Rj[k] = Ro[k];
    }
    Pose& me = link[j];
    mat_mul(par.R, Rj, me.R);
    const double* o = m.origin_xyz + j * 3;
#pragma unroll
    for (int r = 0; r < 3; ++r) me.p[r] = par.p[r] + ((par.R[r * 3] * o[0] + par.R[r * 3 + 1] * o[1]) + par.R[r * 3 + 2] * o[2]);
  }
  for (int b = 0; b < m.n_bodies; ++b) {
    const Pose& ps = m.body_joint[b] < 0 ? root : link[m.body_joint[b]];
    double q[4];
    matrix_to_quat(ps.R, q);
    const int64_t o = f * m.n_bodies + b;
    body_pos[o * 3] = (float)ps.p[0]; body_pos[o * 3 + 1] = (float)ps.p[1]; body_pos[o * 3 + 2] = (float)ps.p[2];
    body_rot[o * 4] = (float)q[0]; body_rot[o * 4 + 1] = (float)q[1]; body_rot[o * 4 + 2] = (float)q[2]; body_rot[o * 4 + 3] = (float)q[3];
  }
}

// ---- finite differences + Gaussian smoothing along the frame axis of a [N, W] array ------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void convert_diff_kernel(const T* __restrict__ x, int64_t n, int w, double dt, T* __restrict__ v) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n * w) return;
  const int64_t i = e / w;
  // numpy: (x[i+1] - x[i-1]) / (2 * dt) with the Python-float divisor cast to the array's type
  if (n == 1) v[e] = (T)0;
  else if (i == 0) v[e] = (x[e + w] - x[e]) / (T)dt;
  else if (i == n - 1) v[e] = (x[e] - x[e - w]) / (T)dt;
  else v[e] = (x[e + w] - x[e - w]) / (T)(2.0 * dt);
}

struct GaussWeights { double w[kGaussRadius + 1]; };  // w[0] centre ... w[4] offset 4 (already normalised)

template <typename T>
__global__ __launch_bounds__(kBlock) void convert_gauss_kernel(const T* __restrict__ x, int64_t n, int w, GaussWeights gw, T* __restrict__ y) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n * w) return;
  const int64_t i = e / w;
  const int c = (int)(e - i * w);
  auto at = [&](int64_t k) -> double {  // 'reflect': d c b a | a b c d | d c b a
    const int64_t period = 2 * n;
    k %= period;
    if (k < 0) k += period;
    if (k >= n) k = period - 1 - k;
    return (double)x[k * w + c];
  };
  // scipy correlate1d, symmetric weights: centre first, then pairs from the outermost offset inwards
  double acc = at(i) * gw.w[0];
  for (int j = kGaussRadius; j >= 1; --j) acc += (at(i - j) + at(i + j)) * gw.w[j];
  y[e] = (T)acc;
}

// ---- angular velocity ----------------------------------------------------------------------------------------------
// compute_angular_velocity (data_convert.py:85-109) on float32 (w, x, y, z) quaternions; `np1`: the scalar promotion
// of numpy < 2 (w widened to double before clip / arccos / sqrt), else everything in float32
__device__ __forceinline__ void angular_velocity_pair(const float* qp, const float* qn, double dt, int np1, float* out) {
  out[0] = out[1] = out[2] = 0.0f;
  const float w = qp[0], x = qp[1], y = qp[2], z = qp[3];
  float n2 = ((w * w + x * x) + y * y) + z * z;
  if (n2 < 1e-8f) n2 = 1e-8f;
  const float w1 = w / n2, x1 = -x / n2, y1 = -y / n2, z1 = -z / n2;
  const float w2 = qn[0], x2 = qn[1], y2 = qn[2], z2 = qn[3];
  float r[4];
  r[0] = ((w1 * w2 - x1 * x2) - y1 * y2) - z1 * z2;
  r[1] = ((w1 * x2 + x1 * w2) + y1 * z2) - z1 * y2;
  r[2] = ((w1 * y2 - x1 * z2) + y1 * w2) + z1 * x2;
  r[3] = ((w1 * z2 + x1 * y2) - y1 * x2) + z1 * w2;
  const float nrm = sqrtf(((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]) + r[3] * r[3]);
  if ((double)nrm < 1e-8) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] /= nrm;
  if (r[0] < 0.0f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = -r[k];
  }
  if (np1) {
    double wc = (double)r[0];
    wc = wc < -1.0 ? -1.0 : (wc > 1.0 ? 1.0 : wc);
    const double angle = 2.0 * acos(wc);
    const double sh = sqrt(1.0 - wc * wc);
    if (sh < 1e-8) return;
    const float shf = (float)sh, sc = (float)(angle / dt);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = sc * (r[1 + k] / shf);
  } else {
    const float wc = fminf(fmaxf(r[0], -1.0f), 1.0f);
    const float angle = 2.0f * acosf(wc);
    const float sh = sqrtf(1.0f - wc * wc);
    if ((double)sh < 1e-8) return;
    const float sc = angle / (float)dt;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = sc * (r[1 + k] / sh);
  }
}

__global__ __launch_bounds__(kBlock) void convert_angvel_kernel(const float* __restrict__ rot, int64_t n, int n_bodies, double dt,
                                                                int np1, float* __restrict__ av) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n * n_bodies) return;
  const int64_t k = e / n_bodies;
  const int64_t stride = (int64_t)n_bodies * 4;
  const float* q = rot + e * 4;
  float o[3] = {0.0f, 0.0f, 0.0f};
  if (n > 1) {
    if (k == 0) {
      angular_velocity_pair(q, q + stride, dt, np1, o);
    } else if (k == n - 1) {
      angular_velocity_pair(q - stride, q, dt, np1, o);
    } else {
      float a[3], b[3];
      angular_velocity_pair(q - stride, q, dt, np1, a);
      angular_velocity_pair(q, q + stride, dt, np1, b);
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = 0.5f * (a[c] + b[c]);
    }
  }
  av[e * 3] = o[0]; av[e * 3 + 1] = o[1]; av[e * 3 + 2] = o[2];
}

template <typename T>
static int diff_then_gauss(const T* x, int64_t n, int w, double dt, const GaussWeights& gw, T* tmp, T* out, hipStream_t st) {
  const unsigned grid = (unsigned)((n * w + kBlock - 1) / kBlock);
  {
    amp::TraceScope trace__("convert_diff_kernel", st);
    convert_diff_kernel<T><<<grid, kBlock, 0, st>>>(x, n, w, dt, tmp);
  }
  int rc = launch_status("convert_diff_kernel");
  if (rc != AMP_OK) return rc;
  {
    amp::TraceScope trace__("convert_gauss_kernel", st);
    convert_gauss_kernel<T><<<grid, kBlock, 0, st>>>(tmp, n, w, gw, out);
  }
  return launch_status("convert_gauss_kernel");
}

}  // namespace amp

using namespace amp;

extern "C" {

int amp_converter_destroy(AmpConverter* c) {
  if (!c) return AMP_OK;
  (void)hipFree(c->parent);
  (void)hipFree(c->qidx);
  (void)hipFree(c->origin_rot);
  (void)hipFree(c->origin_xyz);
  (void)hipFree(c->axis);
  (void)hipFree(c->body_joint);
  delete c;
  return AMP_OK;
}

int amp_converter_create(const AmpKinModel* m, AmpConverter** out) {
  AMP_REQUIRE(m && out, "amp_converter_create: null argument");
  AMP_REQUIRE(m->n_joints >= 0 && m->n_joints <= kMaxJoints, "amp_converter_create: n_joints=%d outside [0, %d]", m->n_joints, kMaxJoints);
  AMP_REQUIRE(m->n_dof >= 1 && m->n_bodies >= 1, "amp_converter_create: empty model");
  AMP_REQUIRE(m->n_joints == 0 || (m->parent && m->qidx && m->origin_rot && m->origin_xyz && m->axis), "amp_converter_create: null joint array");
  AMP_REQUIRE(m->body_joint, "amp_converter_create: null body_joint");
  for (int j = 0; j < m->n_joints; ++j) {
    AMP_REQUIRE(m->parent[j] >= -1 && m->parent[j] < j, "amp_converter_create: joint %d: parent %d is not an earlier joint", j, m->parent[j]);
    AMP_REQUIRE(m->qidx[j] >= -1 && m->qidx[j] < m->n_dof, "amp_converter_create: joint %d: angle column %d out of range", j, m->qidx[j]);
  }
  for (int b = 0; b < m->n_bodies; ++b)
    AMP_REQUIRE(m->body_joint[b] >= -1 && m->body_joint[b] < m->n_joints, "amp_converter_create: body %d: joint %d out of range", b, m->body_joint[b]);
  AmpConverter* c = new (std::nothrow) AmpConverter();
  AMP_REQUIRE(c, "amp_converter_create: out of host memory");
  *c = AmpConverter{};
  c->n_joints = m->n_joints; c->n_dof = m->n_dof; c->n_bodies = m->n_bodies;
  const size_t J = (size_t)(m->n_joints > 0 ? m->n_joints : 1);
  hipError_t e = hipMalloc(&c->parent, sizeof(int32_t) * J);
  if (e == hipSuccess) e = hipMalloc(&c->qidx, sizeof(int32_t) * J);
  if (e == hipSuccess) e = hipMalloc(&c->origin_rot, sizeof(double) * 9 * J);
  if (e == hipSuccess) e = hipMalloc(&c->origin_xyz, sizeof(double) * 3 * J);
  if (e == hipSuccess) e = hipMalloc(&c->axis, sizeof(double) * 3 * J);
  if (e == hipSuccess) e = hipMalloc(&c->body_joint, sizeof(int32_t) * m->n_bodies);
  if (e == hipSuccess && m->n_joints) e = hipMemcpy(c->parent, m->parent, sizeof(int32_t) * m->n_joints, hipMemcpyHostToDevice);
  if (e == hipSuccess && m->n_joints) e = hipMemcpy(c->qidx, m->qidx, sizeof(int32_t) * m->n_joints, hipMemcpyHostToDevice);
  if (e == hipSuccess && m->n_joints) e = hipMemcpy(c->origin_rot, m->origin_rot, sizeof(double) * 9 * m->n_joints, hipMemcpyHostToDevice);
  if (e == hipSuccess && m->n_joints) e = hipMemcpy(c->origin_xyz, m->origin_xyz, sizeof(double) * 3 * m->n_joints, hipMemcpyHostToDevice);
  if (e == hipSuccess && m->n_joints) e = hipMemcpy(c->axis, m->axis, sizeof(double) * 3 * m->n_joints, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->body_joint, m->body_joint, sizeof(int32_t) * m->n_bodies, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    amp_converter_destroy(c);
    return fail(AMP_ERR_HIP, "amp_converter_create: %s", hipGetErrorString(e));
  }
  *out = c;
  return AMP_OK;
}

int64_t amp_convert_workspace_bytes(const AmpConverter* c, int64_t n_rows) {
  if (!c || n_rows < 1) return -1;
  const int64_t n = 2 * n_rows - 1;
  // root_pos [n,3] f64, root_quat [n,4] f64, dof tmp [n,D] f64, body tmp [n,B,3] f32 x 2
  return 8 * n * 7 + 8 * n * c->n_dof + 2 * 4 * n * c->n_bodies * 3 + 256;
}

int amp_convert_motion(const AmpConverter* c, const float* csv_dev, int64_t n_rows, int32_t n_cols, int32_t fps_out,
                       int32_t numpy1_promotion, const AmpConvertOutputs* o, void* workspace_dev, amp_stream_t stream) {
  AMP_REQUIRE(c && o, "amp_convert_motion: null argument");
  AMP_REQUIRE(n_rows >= 2, "amp_convert_motion: at least two CSV rows are needed, got %lld", (long long)n_rows);
  AMP_REQUIRE(n_cols == 7 + c->n_dof, "amp_convert_motion: the CSV has %d columns, the model needs 7 + %d", n_cols, c->n_dof);
  AMP_REQUIRE(fps_out >= 1, "amp_convert_motion: fps must be positive");
  AMP_REQUIRE(csv_dev && workspace_dev, "amp_convert_motion: null buffer");
  AMP_REQUIRE(o->dof_positions && o->dof_velocities && o->body_positions && o->body_rotations && o->body_linear_velocities &&
                  o->body_angular_velocities, "amp_convert_motion: null output");
  AMP_REQUIRE((uintptr_t)workspace_dev % 8 == 0, "amp_convert_motion: workspace must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = 2 * n_rows - 1;
  const int D = c->n_dof, B = c->n_bodies;
  const double dt = 1.0 / (double)fps_out;
  double* root_pos = (double*)workspace_dev;
  double* root_quat = root_pos + n * 3;
  double* dof_tmp = root_quat + n * 4;
  float* body_tmp = (float*)(dof_tmp + n * D);
  float* body_tmp2 = body_tmp + n * B * 3;
  GaussWeights gw;
  {
    // scipy _gaussian_kernel1d(sigma = 1, radius = 4): exp(-0.5 x^2) / sum over x = -4..4 in that order
    double phi[2 * kGaussRadius + 1];
    for (int x = -kGaussRadius; x <= kGaussRadius; ++x) phi[x + kGaussRadius] = exp(-0.5 * (double)(x * x));
    // numpy's pairwise sum of 9 doubles: eight running lanes, the ninth value joins lane 0, then a balanced tree
    const double sum = (((phi[0] + phi[8]) + phi[1]) + (phi[2] + phi[3])) + ((phi[4] + phi[5]) + (phi[6] + phi[7]));
    for (int j = 0; j <= kGaussRadius; ++j) gw.w[j] = phi[kGaussRadius + j] / sum;
  }
  {
    amp::TraceScope trace__("convert_upsample_kernel", st);
    convert_upsample_kernel<<<(unsigned)((n * n_cols + kBlock - 1) / kBlock), kBlock, 0, st>>>(csv_dev, n_rows, n_cols, root_pos, root_quat,
                                                                                          o->dof_positions);
  }
  int rc = launch_status("convert_upsample_kernel");
  if (rc != AMP_OK) return rc;
  {
    amp::TraceScope trace__("convert_fk_kernel", st);
    convert_fk_kernel<<<(unsigned)((n + 63) / 64), 64, 0, st>>>(*c, n, root_pos, root_quat, o->dof_positions, o->body_positions,
                                                              o->body_rotations);
  }
  rc = launch_status("convert_fk_kernel");
  if (rc != AMP_OK) return rc;
  rc = diff_then_gauss<double>(o->dof_positions, n, D, dt, gw, dof_tmp, o->dof_velocities, st);
  if (rc != AMP_OK) return rc;
  rc = diff_then_gauss<float>(o->body_positions, n, B * 3, dt, gw, body_tmp, o->body_linear_velocities, st);
  if (rc != AMP_OK) return rc;
  {
    amp::TraceScope trace__("convert_angvel_kernel", st);
    convert_angvel_kernel<<<(unsigned)((n * B + kBlock - 1) / kBlock), kBlock, 0, st>>>(o->body_rotations, n, B, dt, numpy1_promotion, body_tmp2);
  }
  rc = launch_status("convert_angvel_kernel");
  if (rc != AMP_OK) return rc;
  {
    amp::TraceScope trace__("convert_gauss_kernel", st);
    convert_gauss_kernel<float><<<(unsigned)((n * B * 3 + kBlock - 1) / kBlock), kBlock, 0, st>>>(body_tmp2, n, B * 3, gw, o->body_angular_velocities);
  }
  return launch_status("convert_gauss_kernel");
}

}  // extern "C"
