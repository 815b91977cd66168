"""Helpers shared by the CPU (oracle) and GPU (HIP) parity tests: fixture loading + derived inputs."""

from __future__ import annotations

import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
MOTIONS = os.path.join(os.path.dirname(HERE), "humanoid_amp_amd", "motions")

G1_KEY_BODIES = ["right_rubber_hand", "left_rubber_hand", "right_ankle_roll_link", "left_ankle_roll_link"]
HUM_KEY_BODIES = ["right_hand", "left_hand", "right_foot", "left_foot"]

CLIPSETS = {
    "g1_walk": ["G1_walk"],
    "g1_dance": ["G1_dance"],
    "humanoid3": ["humanoid_walk", "humanoid_run", "humanoid_dance"],
}


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def clip_files(tag: str):
    return [os.path.join(MOTIONS, n + ".npz") for n in CLIPSETS[tag]]


def g1_robot_names():
    m = golden("meta")
    return m["g1_robot_joint_names"].tolist(), m["g1_robot_body_names"].tolist()


def obs_inputs(fx, step: int, prev_amp, prev_hist=None):
    """Rebuild the inputs of the ``_get_observations`` call of ``step`` (see gen_golden.py): the sim state is
    ``in_*`` with the reset rows overwritten, the AMP buffer is the previous output with the reset rows
    replaced by the reference-state expert rows."""
    p = f"s{step}_"
    ref = int(fx["ref_body_index"])
    st = {k: fx[p + "in_" + k].copy() for k in ("joint_pos", "joint_vel", "body_pos_w", "body_quat_w",
                                                  "body_lin_vel_w", "body_ang_vel_w")}
    amp = prev_amp.copy()
    ids = fx[p + "out_reset_env_ids"]
    if len(ids) > 0 and (p + "out_reset_root_state") in fx.files:
        root = fx[p + "out_reset_root_state"]
        st["joint_pos"][ids] = fx[p + "out_reset_dof_pos"]
        st["joint_vel"][ids] = fx[p + "out_reset_dof_vel"]
        st["body_pos_w"][ids, ref] = root[:, 0:3]
        st["body_quat_w"][ids, ref] = root[:, 3:7]
        st["body_lin_vel_w"][ids, ref] = root[:, 7:10]
        st["body_ang_vel_w"][ids, ref] = root[:, 10:13]
        amp[ids] = fx[p + "out_reset_amp_rows"]
    return st, amp
