"""Oracle: CSV -> npz motion converter (TEST INFRASTRUCTURE ONLY).

CPU restatement of motions/data_convert.py of the reference: 30 -> 60 fps up-sampling (:204-232), forward kinematics
(:327-356), central-difference + Gaussian-smoothed velocities (:284-289, :358-365) and quaternion-difference angular
velocities (:85-109, :367-383).  It calls the same scipy routines the reference calls (interp1d, Rotation/Slerp,
gaussian_filter1d); forward kinematics is our own chain product over the JSON model extracted from the URDF, because
Pinocchio is absent here.  PINNED by the reference's own output: motions/custom_motion.npz is data_convert.py run on
datasets/walk1_subject1.csv rows [110:265] (tests/golden/convert_*.npz, tests/test_oracle_convert.py).
"""

from __future__ import annotations

import json

import numpy as np
from scipy.interpolate import interp1d
from scipy.ndimage import gaussian_filter1d
from scipy.spatial.transform import Rotation, Slerp

FPS_IN = 30  # the LAFAN1 retargeting CSVs (data_convert.py:197)


def load_model(path: str) -> dict:
    with open(path) as fh:
        return json.load(fh)


def rpy_matrix(rpy) -> np.ndarray:
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def axis_angle_matrix(axis, angle: float) -> np.ndarray:
    """Rodrigues rotation about a unit axis (what a revolute joint contributes)."""
    x, y, z = axis
    c, s = np.cos(angle), np.sin(angle)
    t = 1.0 - c
    return np.array([[t * x * x + c, t * x * y - s * z, t * x * z + s * y],
                     [t * x * y + s * z, t * y * y + c, t * y * z - s * x],
                     [t * x * z - s * y, t * y * z + s * x, t * z * z + c]])


def matrix_to_quat_wxyz(m: np.ndarray) -> np.ndarray:
    """Rotation matrix -> (w, x, y, z) with Eigen's branch structure (what pin.Quaternion(R) runs, :348-352): the sign
    convention of the stored quaternions follows from it."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0.0:
        t = np.sqrt(t + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (m[2, 1] - m[1, 2]) * t
        q[2] = (m[0, 2] - m[2, 0]) * t
        q[3] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[1 + i] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[k, j] - m[j, k]) * t
        q[1 + j] = (m[j, i] + m[i, j]) * t
        q[1 + k] = (m[k, i] + m[i, k]) * t
    return q


def topo_joints(model: dict):
    """Joints ordered parent-before-child, with the index of the parent LINK's joint (-1 = root link)."""
    by_child = {j["child"]: j for j in model["joints"]}
    order, seen = [], set(model["root_links"])
    pending = list(model["joints"])
    while pending:
        rest = []
        for j in pending:
            if j["parent"] in seen:
                order.append(j)
                seen.add(j["child"])
            else:
                rest.append(j)
        if len(rest) == len(pending):
            raise ValueError("kinematic tree is not connected")
        pending = rest
    index = {j["child"]: i for i, j in enumerate(order)}
    parents = [index.get(j["parent"], -1) for j in order]
    return order, parents, by_child


def forward_kinematics(model: dict, root_pos, root_quat_xyzw, joint_pos, joint_names, body_names):
    """World pose of every named link for ONE frame (fp64): (positions [B,3], rotation matrices [B,3,3])."""
    order, parents, _ = topo_joints(model)
    qidx = {n: i for i, n in enumerate(joint_names)}
    R0 = Rotation.from_quat(root_quat_xyzw).as_matrix()
    p0 = np.asarray(root_pos, dtype=np.float64)
    Rw, pw = [], []
    for j, par in zip(order, parents):
        Rp, pp = (R0, p0) if par < 0 else (Rw[par], pw[par])
        Rj = rpy_matrix(j["rpy"])
        if j["type"] == "revolute":
            Rj = Rj @ axis_angle_matrix(j["axis"], float(joint_pos[qidx[j["name"]]]))
        Rw.append(Rp @ Rj)
        pw.append(pp + Rp @ np.asarray(j["xyz"]))
    link = {j["child"]: i for i, j in enumerate(order)}
    pos = np.zeros((len(body_names), 3))
    rot = np.zeros((len(body_names), 3, 3))
    for b, name in enumerate(body_names):
        if name in model["root_links"]:
            pos[b], rot[b] = p0, R0
        else:
            pos[b], rot[b] = pw[link[name]], Rw[link[name]]
    return pos, rot


def angular_velocity(q_prev, q_next, dt, promotion="numpy2", eps=1e-8):
    """Restates compute_angular_velocity (:85-109) on float32 wxyz quaternions.  The function mixes float32 numpy
    scalars with Python floats, so its arithmetic depends on the numpy generation it ran under, and the reference ships
    files from both:
      promotion="numpy2" (weak Python scalars: everything stays float32) reproduces motions/G1_walk.npz to 5e-7;
      promotion="numpy1" (a float32 SCALAR meeting a Python float becomes float64: w is widened before clip / arccos /
      sqrt; the axis division and final scaling stay float32) reproduces motions/custom_motion.npz bit for bit on 92 %
      of the entries -- the rest are near-identity rotations where one ulp of the float32 w moves the angle by up to
      3.5e-4 rad (0.02 rad/s at 60 fps): the formulation's own resolution, not reproducible across BLAS builds."""
    w, x, y, z = q_prev
    n2 = w * w + x * x + y * y + z * z
    if n2 < 1e-8:
        n2 = np.float32(1e-8)
    qi = np.array([w, -x, -y, -z], dtype=np.float32) / n2
    w1, x1, y1, z1 = qi
    w2, x2, y2, z2 = q_next
    rel = np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                    w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dtype=np.float32)
    nrm = np.linalg.norm(rel)  # float32 BLAS dot: its summation order is the host library's (see the docstring)
    if nrm < eps:
        return np.zeros(3, dtype=np.float32)
    rel = rel / nrm
    if rel[0] < 0.0:
        rel = -rel
    if promotion == "numpy2":
        wc = np.clip(rel[0], np.float32(-1.0), np.float32(1.0))
        angle = np.float32(2.0) * np.arccos(wc)
        sin_half = np.sqrt(np.float32(1.0) - wc * wc)
        if sin_half < eps:
            return np.zeros(3, dtype=np.float32)
        return (angle / np.float32(dt)) * (rel[1:] / sin_half)
    wc = min(max(np.float64(rel[0]), -1.0), 1.0)
    angle = 2.0 * np.arccos(wc)
    sin_half = np.sqrt(1.0 - wc * wc)
    if sin_half < eps:
        return np.zeros(3, dtype=np.float32)
    axis = rel[1:] / np.float32(sin_half)
    return np.float32(angle / dt) * axis


def convert(csv_rows: np.ndarray, model: dict, joint_names, body_names, fps: int = 60, promotion: str = "numpy2") -> dict:
    """csv_rows [N0, 7 + D] float32 (root xyz, root quat xyzw, D joints) -> the npz arrays of data_convert.main()."""
    data = np.asarray(csv_rows, dtype=np.float32)
    n0 = data.shape[0]
    t0 = np.linspace(0, (n0 - 1) * (1.0 / FPS_IN), n0)
    n = 2 * n0 - 1
    t1 = np.linspace(0, (n0 - 1) * (1.0 / FPS_IN), n)
    dt = 1.0 / fps
    root_pos = interp1d(t0, data[:, 0:3], axis=0, kind="linear")(t1)
    root_quat = Slerp(t0, Rotation.from_quat(data[:, 3:7]))(t1).as_quat()
    joints = interp1d(t0, data[:, 7:], axis=0, kind="linear")(t1)

    def velocity(x):
        v = np.zeros_like(x)
        v[1:-1] = (x[2:] - x[:-2]) / (2 * dt)
        v[0] = (x[1] - x[0]) / dt
        v[-1] = (x[-1] - x[-2]) / dt
        return v

    dof_vel = gaussian_filter1d(velocity(joints), sigma=1, axis=0)
    B = len(body_names)
    body_pos = np.zeros((n, B, 3), dtype=np.float32)
    body_rot = np.zeros((n, B, 4), dtype=np.float32)
    for i in range(n):
        pos, rot = forward_kinematics(model, root_pos[i], root_quat[i], joints[i], joint_names, body_names)
        body_pos[i] = pos
        for b in range(B):
            body_rot[i, b] = matrix_to_quat_wxyz(rot[b])
    body_lin = gaussian_filter1d(velocity(body_pos), sigma=1, axis=0)
    body_ang = np.zeros((n, B, 3), dtype=np.float32)
    for b in range(B):
        q = body_rot[:, b, :]
        av = np.zeros((n, 3), dtype=np.float32)
        if n > 1:
            av[0] = angular_velocity(q[0], q[1], dt, promotion)
            av[-1] = angular_velocity(q[-2], q[-1], dt, promotion)
        for k in range(1, n - 1):
            av[k] = 0.5 * (angular_velocity(q[k - 1], q[k], dt, promotion) + angular_velocity(q[k], q[k + 1], dt, promotion))
        body_ang[:, b, :] = gaussian_filter1d(av, sigma=1, axis=0)
    return {"fps": np.int64(fps), "dof_names": np.array(joint_names, dtype=np.str_), "body_names": np.array(body_names, dtype=np.str_),
            "dof_positions": joints, "dof_velocities": dof_vel, "body_positions": body_pos, "body_rotations": body_rot,
            "body_linear_velocities": body_lin, "body_angular_velocities": body_ang}
