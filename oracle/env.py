"""Oracle: AMP feature extraction, K-frame history, dones, reset ids, task reward (TEST INFRASTRUCTURE ONLY).

CPU restatement (torch-CPU fp32, op-for-op) of the env-side hot path of the reference:
``g1_amp_env.py`` / ``humanoid_amp_env.py``.  Pinned by tests/golden/collect_*, envstep_*, rewards_fn.

Third-party pieces restated from published formulae (absent from /root/reference -> their fp32
operation ORDER is "parity unpinned"): ``isaaclab.utils.math.quat_apply`` / ``quat_rotate_inverse``
(Isaac Lab 2.2.0) and ``DirectRLEnv.step``'s ``reset_buf.nonzero()``.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from . import motion as om

KEY_BODY_OBS = 12  # 4 key bodies x 3 (g1_amp_env.py:94)


# --- third-party math, restated (unpinned op order) -------------------------------------------------


def quat_apply(quat: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    """v + w*t + q_xyz x t with t = 2*(q_xyz x v); wxyz.  Call sites g1_amp_env.py:495-496."""
    shape = vec.shape
    quat = quat.reshape(-1, 4)
    vec = vec.reshape(-1, 3)
    xyz = quat[:, 1:]
    t = xyz.cross(vec, dim=-1) * 2
    return (vec + quat[:, 0:1] * t + xyz.cross(t, dim=-1)).view(shape)


def quat_rotate_inverse(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """v(2w^2-1) - 2w(q_v x v) + 2 q_v (q_v . v); wxyz.  Call site g1_amp_env.py:253."""
    q_w = q[..., 0]
    q_vec = q[..., 1:]
    a = v * (2.0 * q_w**2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * torch.bmm(q_vec.view(q.shape[0], 1, 3), v.view(q.shape[0], 3, 1)).squeeze(-1) * 2.0
    return a - b + c


# --- AMP features ----------------------------------------------------------------------------------


def tangent_and_normal(q: torch.Tensor) -> torch.Tensor:
    """R(q) e_x | R(q) e_z  (g1_amp_env.py:489-497)."""
    ex = torch.zeros_like(q[..., :3])
    ez = torch.zeros_like(q[..., :3])
    ex[..., 0] = 1
    ez[..., -1] = 1
    return torch.cat([quat_apply(q, ex), quat_apply(q, ez)], dim=-1)


def compute_obs(dof_pos, dof_vel, root_pos, root_rot, root_lin, root_ang, key_pos) -> torch.Tensor:
    """[q | qd | z | tangent | normal | v (world) | w (world) | key - root]  (g1_amp_env.py:535-561)."""
    rel = (key_pos - root_pos.unsqueeze(-2)).view(key_pos.shape[0], -1)
    return torch.cat(
        [dof_pos, dof_vel, root_pos[:, 2:3], tangent_and_normal(root_rot), root_lin, root_ang, rel], dim=-1
    )


def collect_reference(mt: om.MotionTables, times, motion_ids, K: int, dof_perm, ref_body: int, key_bodies):
    """Expert AMP observations [n, K*D] for explicit (times, ids)  (g1_amp_env.py:445-486)."""
    times = np.asarray(times, dtype=np.float64)
    t_hist = om.history_times(mt, times, K)
    if motion_ids is not None:
        ids = np.repeat(motion_ids, K)
    else:
        ids = np.zeros_like(t_hist, dtype=np.int32)
    dp, dv, bp, br, bl, ba = om.sample(mt, t_hist, ids)
    obs = compute_obs(
        dp[:, dof_perm], dv[:, dof_perm], bp[:, ref_body], br[:, ref_body], bl[:, ref_body], ba[:, ref_body],
        bp[:, key_bodies],
    )
    return obs.view(len(times), -1)


def shift_history(buf: torch.Tensor, obs: torch.Tensor) -> torch.Tensor:
    """In place: slot k+1 <- slot k (k = K-2..0), slot 0 <- obs; returns the [N, K*D] view (g1_amp_env.py:187-193)."""
    K = buf.shape[1]
    for i in reversed(range(K - 1)):
        buf[:, i + 1] = buf[:, i]
    buf[:, 0] = obs.clone()
    return buf.view(buf.shape[0], -1)


def actor_observation(obs, last_actions, command, *, use_command: bool, n_actor: int = 1, hist_buf=None,
                      just_reset=None, hist_actions: bool = True, hist_command: bool = True) -> torch.Tensor:
    """Policy observation incl. the optional actor history with reset warm-start (g1_amp_env.py:195-242)."""
    base = obs[:, :-KEY_BODY_OBS]
    if n_actor <= 1:
        out = torch.cat([base, last_actions], dim=-1)
        if use_command:
            out = torch.cat([out, command], dim=-1)
        return out
    cur = [base, last_actions] + ([command] if use_command else [])
    cur = torch.cat(cur, dim=-1)
    hist = [base]
    if hist_actions:
        hist.append(last_actions)
    if hist_command and use_command:
        hist.append(command)
    hist = torch.cat(hist, dim=-1)
    if just_reset.any():
        for i in range(n_actor - 1):
            hist_buf[just_reset, i] = hist[just_reset]
        just_reset[:] = False
    for i in reversed(range(n_actor - 2)):
        hist_buf[:, i + 1] = hist_buf[:, i]
    hist_buf[:, 0] = hist
    return torch.cat([cur, hist_buf.view(obs.shape[0], -1)], dim=-1)


# --- dones / reset ids -----------------------------------------------------------------------------


def dones(episode_length_buf, max_episode_length: int, root_z, termination_height: float, early_termination=True):
    """(died, time_out)  (g1_amp_env.py:321-330)."""
    time_out = episode_length_buf >= max_episode_length - 1
    died = root_z < termination_height if early_termination else torch.zeros_like(time_out)
    return died, time_out


def reset_env_ids(died: torch.Tensor, time_out: torch.Tensor) -> torch.Tensor:
    """Ascending int64 ids of (terminated | time_out) -- DirectRLEnv.step [restated, third-party]."""
    return (died | time_out).nonzero(as_tuple=False).squeeze(-1)


# --- task reward -----------------------------------------------------------------------------------


def exp_reward_with_floor(error: torch.Tensor, weight: float, sigma: float, floor: float = 3.0) -> torch.Tensor:
    """Exponential inside floor*sigma^2, C1-continuous linear outside (g1_amp_env.py:500-532).

    In the reference (TorchScript) ``torch.exp(-floor)`` acts on a python float, i.e. the three scalars
    are float64 host values; only the tensor ops round to fp32 (the scalar operand is cast to fp32).
    """
    sigma_sq = sigma * sigma
    threshold = floor * sigma_sq
    val_at_thr = weight * math.exp(-floor)
    slope = weight / sigma_sq * math.exp(-floor)
    linear = val_at_thr - slope * (error - threshold)
    expo = weight * torch.exp(-error / sigma_sq)
    return torch.where(error > threshold, linear, expo)


def compute_rewards(s_term, s_act, s_lim, s_acc, s_vel, terminated, actions, joint_pos, limits, joint_acc, joint_vel):
    """Sum of the five penalty terms + their per-term tensors (g1_amp_env.py:564-606)."""
    r_term = s_term * terminated.float()
    r_act = s_act * torch.sum(torch.square(actions), dim=1)
    out = -(joint_pos - limits[:, :, 0]).clip(max=0.0)
    out += (joint_pos - limits[:, :, 1]).clip(min=0.0)
    r_lim = s_lim * torch.sum(out, dim=1)
    r_acc = s_acc * torch.sum(torch.square(joint_acc), dim=1)
    r_vel = s_vel * torch.sum(torch.square(joint_vel), dim=1)
    total = r_term + r_act + r_lim + r_acc + r_vel
    return total, dict(pub_termination=r_term, pub_action_l2=r_act, pub_joint_pos_limits=r_lim,
                       pub_joint_acc_l2=r_acc, pub_joint_vel_l2=r_vel)


def g1_task_reward(cfg, root_lin_w, root_quat_w, command, terminated, actions, joint_pos, limits, joint_acc, joint_vel):
    """Total task reward + the tensors behind the logged means (g1_amp_env.py:246-319)."""
    parts = {}
    if cfg["rew_track_vel"] > 0.0:
        v_b = quat_rotate_inverse(root_quat_w, root_lin_w)[:, :2]
        err = torch.norm(v_b - command, dim=-1)
        track = exp_reward_with_floor(torch.square(err), cfg["rew_track_vel"], 0.5, floor=4.0)
        parts["rew_track_vel"], parts["error_track_vel"] = track, err
    else:
        track = torch.zeros(root_lin_w.shape[0], dtype=torch.float)
    basic, log = compute_rewards(cfg["rew_termination"], cfg["rew_action_l2"], cfg["rew_joint_pos_limits"],
                                 cfg["rew_joint_acc_l2"], cfg["rew_joint_vel_l2"], terminated, actions, joint_pos,
                                 limits, joint_acc, joint_vel)
    total = basic + track
    parts.update(log)
    parts["total_reward"] = total
    return total, parts


# --- reference-state initialisation on reset ---------------------------------------------------------


def reset_reference_state(mt: om.MotionTables, times, motion_ids, dof_perm, root_body: int, env_origins, z_lift: float):
    """root_state [n,13] (pos+origin, z lifted; quat; lin; ang), dof_pos, dof_vel (g1_amp_env.py:385-411)."""
    dp, dv, bp, br, bl, ba = om.sample(mt, times, motion_ids)
    root = torch.zeros(len(times), 13)
    root[:, 0:3] = bp[:, root_body] + env_origins
    root[:, 2] += z_lift
    root[:, 3:7] = br[:, root_body]
    root[:, 7:10] = bl[:, root_body]
    root[:, 10:13] = ba[:, root_body]
    return root, dp[:, dof_perm], dv[:, dof_perm]
