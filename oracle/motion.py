"""Oracle: time-indexed LERP / SLERP sampling of reference-motion clips (TEST INFRASTRUCTURE ONLY).

CPU restatement of ``motions/motion_loader.py`` of the reference.  Index math is numpy float64 /
int64 exactly as the reference does it on the host; table math is torch-CPU float32 issued as the
same ATen ops in the same order, so results are bit-identical to the reference on CPU
(tests/test_oracle_golden.py checks that against tests/golden/*.npz).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import numpy as np
import torch

TABLE_KEYS = (
    "dof_positions",
    "dof_velocities",
    "body_positions",
    "body_rotations",
    "body_linear_velocities",
    "body_angular_velocities",
)


@dataclass
class MotionTables:
    """Concatenated clip tables + per-clip bookkeeping (reference: motion_loader.py:98-164)."""

    dt: np.float64
    traj_starts: np.ndarray  # int64 [C] global index of each clip's first frame
    traj_ends: np.ndarray  # int64 [C] global index of each clip's LAST frame (inclusive)
    durations: np.ndarray  # float64 [C] = dt * (frames - 1)
    dof_names: list
    body_names: list
    tables: dict  # name -> torch.float32 CPU tensor

    @property
    def num_frames(self) -> int:
        return int(self.traj_ends[-1]) + 1

    @property
    def num_trajectories(self) -> int:
        return len(self.traj_starts)


def load_tables(files: Sequence[str]) -> MotionTables:
    """motion_loader.py:116-158 -- dt and names from the FIRST file only; float64 arrays cast to fp32."""
    starts, ends, durs = [], [], []
    parts = {k: [] for k in TABLE_KEYS}
    dt = None
    names = None
    cursor = 0
    for path in files:
        with np.load(path) as npz:
            if dt is None:
                dt = 1.0 / npz["fps"]  # numpy int64 scalar -> float64 (motion_loader.py:122)
                names = (npz["dof_names"].tolist(), npz["body_names"].tolist())
            for k in TABLE_KEYS:
                parts[k].append(npz[k])
            n = npz["dof_positions"].shape[0]
        starts.append(cursor)
        cursor += n
        ends.append(cursor - 1)
        durs.append(dt * (n - 1))
    tables = {k: torch.tensor(np.concatenate(v), dtype=torch.float32) for k, v in parts.items()}
    return MotionTables(
        dt=np.float64(dt),
        traj_starts=np.array(starts),
        traj_ends=np.array(ends),
        durations=np.array(durs),
        dof_names=names[0],
        body_names=names[1],
        tables=tables,
    )


def frame_blend(mt: MotionTables, times: np.ndarray, motion_ids: np.ndarray):
    """(t, clip) -> (i0, i1, blend) in float64/int64  (motion_loader.py:281-307).

    ``round`` (half-to-even), not floor: blend is signed; negative times clamp the index, not the blend.
    """
    dur = mt.durations[motion_ids]
    first = mt.traj_starts[motion_ids]
    span = mt.traj_ends[motion_ids] - first
    phase = np.clip(times / dur, 0.0, 1.0)
    local0 = (phase * span).round(decimals=0).astype(int)
    local1 = np.minimum(local0 + 1, span)
    blend = ((times - local0 * mt.dt) / mt.dt).round(decimals=5)
    return first + local0, first + local1, blend


def lerp_rows(table: torch.Tensor, i0: np.ndarray, i1: np.ndarray, blend: torch.Tensor) -> torch.Tensor:
    """(1-b)*T[i0] + b*T[i1], b broadcast over trailing dims (motion_loader.py:209-215)."""
    a, b = table[i0], table[i1]
    w = blend
    for _ in range(a.ndim - 1):
        w = w.unsqueeze(-1)
    return (1.0 - w) * a + w * b


def slerp_rows(table: torch.Tensor, i0: np.ndarray, i1: np.ndarray, blend: torch.Tensor) -> torch.Tensor:
    """wxyz SLERP with the reference's exact branch structure (motion_loader.py:240-279, Appendix A.3).

    No renormalisation; |sin| < 1e-3 -> plain average (blend ignored); |cos| >= 1 -> q0 (applied last).
    """
    q0, q1 = table[i0], table[i1]
    w = blend
    for _ in range(q0.ndim - 1):
        w = w.unsqueeze(-1)
    cos_h = q0[..., 0] * q1[..., 0] + q0[..., 1] * q1[..., 1] + q0[..., 2] * q1[..., 2] + q0[..., 3] * q1[..., 3]
    flip = cos_h < 0
    q1 = q1.clone()
    q1[flip] = -q1[flip]
    cos_h = torch.abs(cos_h).unsqueeze(-1)
    half = torch.acos(cos_h)
    sin_h = torch.sqrt(1.0 - cos_h * cos_h)
    ra = torch.sin((1 - w) * half) / sin_h
    rb = torch.sin(w * half) / sin_h
    comps = [ra * q0[..., c : c + 1] + rb * q1[..., c : c + 1] for c in (0, 1, 2, 3)]
    out = torch.cat(comps, dim=-1)
    out = torch.where(torch.abs(sin_h) < 0.001, 0.5 * q0 + 0.5 * q1, out)
    out = torch.where(torch.abs(cos_h) >= 1, q0, out)
    return out


def sample(mt: MotionTables, times: np.ndarray, motion_ids: np.ndarray | None = None):
    """The 6-tuple of MotionLoader.sample for explicit times (motion_loader.py:361-390).

    ``motion_ids=None`` -> clip 0 for every sample (motion_loader.py:365-366).
    """
    times = np.asarray(times, dtype=np.float64)
    if motion_ids is None:
        motion_ids = np.zeros(len(times), dtype=np.int32)
    i0, i1, blend64 = frame_blend(mt, times, motion_ids)
    blend = torch.tensor(blend64, dtype=torch.float32)
    t = mt.tables
    return (
        lerp_rows(t["dof_positions"], i0, i1, blend),
        lerp_rows(t["dof_velocities"], i0, i1, blend),
        lerp_rows(t["body_positions"], i0, i1, blend),
        slerp_rows(t["body_rotations"], i0, i1, blend),
        lerp_rows(t["body_linear_velocities"], i0, i1, blend),
        lerp_rows(t["body_angular_velocities"], i0, i1, blend),
    )


def history_times(mt: MotionTables, current_times: np.ndarray, num_amp_observations: int) -> np.ndarray:
    """t - dt*[0..K-1], env-major / newest first, flattened (g1_amp_env.py:454-457)."""
    return (np.expand_dims(current_times, axis=-1) - mt.dt * np.arange(0, num_amp_observations)).flatten()
