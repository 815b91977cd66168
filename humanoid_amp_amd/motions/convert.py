"""CSV -> npz motion converter on the MI355X.

Device counterpart of the reference's offline ``motions/data_convert.py`` (LAFAN1-retargeted CSV at 30 fps: root xyz,
root quaternion xyzw, 29 joint angles -> the 60 fps npz schema of ``motions/README.md:11-21``): same command-line
surface (``--csv --start --end --fps --output``, plus ``--model`` instead of ``--urdf/--meshes``: the kinematic tree is
a small JSON extracted from the URDF by ``tools/urdf_to_kinematics.py``; Pinocchio is not needed), same arrays and
dtypes in the file it writes.  All frames are processed by five batched HIP kernels (``csrc/convert.hip``).

    python -m humanoid_amp_amd.motions.convert --csv datasets/walk1_subject1.csv --start 100 --end 300 --output G1_walk.npz
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
from typing import Optional, Sequence

import numpy as np
import torch

from .. import _native as nat

MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
G1_MODEL = os.path.join(MODELS_DIR, "g1_29dof.json")

# joint order of the LAFAN1-retargeted CSV columns and the links the reference records (data_convert.py:236-318)
G1_CSV_JOINTS = [
    "left_hip_pitch_joint", "left_hip_roll_joint", "left_hip_yaw_joint", "left_knee_joint", "left_ankle_pitch_joint",
    "left_ankle_roll_joint", "right_hip_pitch_joint", "right_hip_roll_joint", "right_hip_yaw_joint", "right_knee_joint",
    "right_ankle_pitch_joint", "right_ankle_roll_joint", "waist_yaw_joint", "waist_roll_joint", "waist_pitch_joint",
    "left_shoulder_pitch_joint", "left_shoulder_roll_joint", "left_shoulder_yaw_joint", "left_elbow_joint",
    "left_wrist_roll_joint", "left_wrist_pitch_joint", "left_wrist_yaw_joint", "right_shoulder_pitch_joint",
    "right_shoulder_roll_joint", "right_shoulder_yaw_joint", "right_elbow_joint", "right_wrist_roll_joint",
    "right_wrist_pitch_joint", "right_wrist_yaw_joint"]
G1_RECORDED_BODIES = [
    "pelvis", "head_link", "torso_link", "left_shoulder_pitch_link", "left_shoulder_roll_link", "left_shoulder_yaw_link",
    "left_elbow_link", "right_shoulder_pitch_link", "right_shoulder_roll_link", "right_shoulder_yaw_link",
    "right_elbow_link", "left_hip_yaw_link", "left_hip_roll_link", "left_hip_pitch_link", "left_knee_link",
    "right_hip_yaw_link", "right_hip_roll_link", "right_hip_pitch_link", "right_knee_link", "right_rubber_hand",
    "left_rubber_hand", "right_ankle_roll_link", "left_ankle_roll_link", "waist_yaw_link", "waist_roll_link"]


def _rpy_matrix(rpy) -> np.ndarray:
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]], dtype=np.float64)


class MotionConverter:
    """``MotionConverter(model_json, joint_names, body_names, device).convert(csv_rows)`` -> dict of numpy arrays."""

    def __init__(self, model_path: str = G1_MODEL, joint_names: Sequence[str] = G1_CSV_JOINTS,
                 body_names: Sequence[str] = G1_RECORDED_BODIES, device="cuda:0"):
        self.device = nat.require_gpu(device)
        self._lib = nat.load()
        self.joint_names, self.body_names = [str(n) for n in joint_names], [str(n) for n in body_names]
        with open(model_path) as fh:
            model = json.load(fh)
        roots = set(model["root_links"])
        if len(roots) != 1:
            raise ValueError(f"the kinematic model must have exactly one root link, found {sorted(roots)}")
        # parent-before-child order
        order, seen, pending = [], set(roots), list(model["joints"])
        while pending:
            rest = [j for j in pending if j["parent"] not in seen]
            ready = [j for j in pending if j["parent"] in seen]
            if not ready:
                raise ValueError("the kinematic tree is not connected")
            for j in ready:
                order.append(j)
                seen.add(j["child"])
            pending = rest
        link_joint = {j["child"]: i for i, j in enumerate(order)}
        qcol = {n: i for i, n in enumerate(self.joint_names)}
        parent = np.array([link_joint.get(j["parent"], -1) for j in order], dtype=np.int32)
        qidx = np.full(len(order), -1, dtype=np.int32)
        for i, j in enumerate(order):
            if j["type"] in ("revolute", "continuous"):
                if j["name"] not in qcol:
                    raise ValueError(f"joint {j['name']} of the model has no CSV column")
                qidx[i] = qcol[j["name"]]
            elif j["type"] != "fixed":
                raise ValueError(f"joint {j['name']}: type {j['type']} is not supported")
        missing = [n for n in self.joint_names if n not in {j["name"] for j in order}]
        if missing:
            raise ValueError(f"CSV joints missing from the model: {missing}")
        rot = np.stack([_rpy_matrix(j["rpy"]) for j in order]).reshape(-1, 9) if order else np.zeros((0, 9))
        xyz = np.array([j["xyz"] for j in order], dtype=np.float64).reshape(-1, 3)
        axis = np.array([j["axis"] for j in order], dtype=np.float64).reshape(-1, 3)
        norm = np.linalg.norm(axis, axis=1, keepdims=True)
        axis = np.where(norm > 0, axis / np.where(norm > 0, norm, 1.0), axis)
        body_joint = np.empty(len(self.body_names), dtype=np.int32)
        for b, name in enumerate(self.body_names):
            if name in roots:
                body_joint[b] = -1
            elif name in link_joint:
                body_joint[b] = link_joint[name]
            else:
                raise ValueError(f"body {name} is not a link of the model")
        self._host = [np.ascontiguousarray(a) for a in (parent, qidx, rot, xyz, axis, body_joint)]
        m = nat.AmpKinModel()
        m.n_joints, m.n_dof, m.n_bodies = len(order), len(self.joint_names), len(self.body_names)
        m.parent, m.qidx, m.origin_rot, m.origin_xyz, m.axis, m.body_joint = (a.ctypes.data for a in self._host)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_converter_create(C.byref(m), C.byref(h)), "amp_converter_create")
        self._handle = h

    def convert(self, csv_rows, fps: int = 60, numpy_generation: int = 2) -> dict:
        """``csv_rows`` [N0, 7 + D] (numpy / tensor; cast to float32 like the reference does) -> the npz arrays.
        ``numpy_generation`` (1 or 2): whose scalar arithmetic the angular-velocity step reproduces (the reference's
        files come from both: G1_walk.npz = 2, custom_motion.npz = 1)."""
        rows = torch.as_tensor(np.asarray(csv_rows, dtype=np.float32)).to(self.device).contiguous()
        n0, cols = rows.shape
        n, D, B = 2 * n0 - 1, len(self.joint_names), len(self.body_names)
        dev = self.device
        out = {"dof_positions": torch.empty((n, D), dtype=torch.float64, device=dev),
               "dof_velocities": torch.empty((n, D), dtype=torch.float64, device=dev),
               "body_positions": torch.empty((n, B, 3), device=dev), "body_rotations": torch.empty((n, B, 4), device=dev),
               "body_linear_velocities": torch.empty((n, B, 3), device=dev),
               "body_angular_velocities": torch.empty((n, B, 3), device=dev)}
        o = nat.AmpConvertOutputs()
        for k, t in out.items():
            setattr(o, k, t.data_ptr())
        ws = torch.empty(max(int(self._lib.amp_convert_workspace_bytes(self._handle, n0)), 8), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            nat.check(self._lib.amp_convert_motion(self._handle, nat.dptr(rows), n0, cols, int(fps), 1 if numpy_generation == 1 else 0,
                                                   C.byref(o), nat.dptr(ws), nat.stream_ptr()), "amp_convert_motion")
            torch.cuda.current_stream().synchronize()
        data = {"fps": np.int64(fps), "dof_names": np.array(self.joint_names, dtype=np.str_),
                "body_names": np.array(self.body_names, dtype=np.str_)}
        data.update({k: t.cpu().numpy() for k, t in out.items()})
        return data

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h is not None and getattr(self, "_lib", None) is not None:
            self._lib.amp_converter_destroy(h)


_SCHEMA = {"dof_positions": 2, "dof_velocities": 2, "body_positions": 3, "body_rotations": 3, "body_linear_velocities": 3,
           "body_angular_velocities": 3}


def save_motion_npz(path: str, data: dict) -> None:
    """Write a clip in the schema ``MotionLoader`` reads (motions/README.md:11-21; the reference's writers:
    data_convert.py:385-399, record_data.py:233-259), checking shapes against the name lists."""
    n = data["dof_positions"].shape[0]
    D, B = len(data["dof_names"]), len(data["body_names"])
    want = {"dof_positions": (n, D), "dof_velocities": (n, D), "body_positions": (n, B, 3), "body_rotations": (n, B, 4),
            "body_linear_velocities": (n, B, 3), "body_angular_velocities": (n, B, 3)}
    for k, shape in want.items():
        if tuple(np.asarray(data[k]).shape) != shape:
            raise ValueError(f"{k} has shape {tuple(np.asarray(data[k]).shape)}, the schema needs {shape}")
    np.savez(path, fps=np.int64(data["fps"]), dof_names=np.asarray(data["dof_names"], dtype=np.str_),
             body_names=np.asarray(data["body_names"], dtype=np.str_), **{k: np.asarray(data[k]) for k in want})


def load_csv(path: str, start: int = 0, end: Optional[int] = None) -> np.ndarray:
    """Rows [start:end] of a header-less CSV as float32 (data_convert.py:182-196)."""
    return np.loadtxt(path, delimiter=",", dtype=np.float32, ndmin=2)[start:end]


def main(argv=None):
    ap = argparse.ArgumentParser(description="Convert a 30 fps motion CSV to the 60 fps npz schema on the MI355X.")
    ap.add_argument("--csv", required=True)
    ap.add_argument("--model", default=G1_MODEL, help="kinematic tree JSON (tools/urdf_to_kinematics.py)")
    ap.add_argument("--output", default="motions/custom_motion.npz")
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--end", type=int, default=None)
    ap.add_argument("--fps", type=int, default=60)
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    rows = load_csv(args.csv, args.start, args.end)
    print(f"Loading CSV: {args.csv}, frames [{args.start}:{args.end}] -> {rows.shape[0]} rows")
    data = MotionConverter(args.model, device=args.device).convert(rows, fps=args.fps)
    save_motion_npz(args.output, data)
    print(f"Conversion completed, data saved to {args.output}: {data['dof_positions'].shape[0]} frames at {args.fps} fps")


if __name__ == "__main__":
    main()
