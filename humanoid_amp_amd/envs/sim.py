"""State provider: what ``DirectRLEnv`` / ``Articulation`` give the reference env, without Isaac Sim.

The reference reads simulator state through ``self.robot.data.*`` and writes resets back through
``self.robot.write_*_to_sim`` (g1_amp_env.py:177-183,348-350).  PhysX is closed and out of scope; this module
keeps the same attribute / method names over plain torch tensors so the env classes run unchanged, and so a real
Isaac Lab ``Articulation`` can be passed in instead (``env = G1AmpEnv(cfg, robot=articulation)``).

``SyntheticArticulation.step`` is a toy integrator (PD joints, ballistic root with a ground clamp), enough to
exercise episode logic; it is test / benchmark scaffolding, not part of the measured path.
"""

from __future__ import annotations

import math
from types import SimpleNamespace

import torch


class SyntheticArticulation:
    def __init__(self, num_envs: int, joint_names, body_names, device, *, root_body: str, dt: float = 1.0 / 60.0,
                 soft_limit: float = 0.9 * math.pi / 2, init_height: float = 0.8, seed: int = 0):
        self.num_envs, self.device, self.dt = int(num_envs), torch.device(device), float(dt)
        N, nd, nb = self.num_envs, len(joint_names), len(body_names)
        f32 = dict(dtype=torch.float32, device=self.device)
        self._gen = torch.Generator(device=self.device).manual_seed(seed)
        self._root = body_names.index(root_body)
        d = SimpleNamespace()
        d.joint_names, d.body_names = list(joint_names), list(body_names)
        d.joint_pos, d.joint_vel, d.joint_acc = (torch.zeros((N, nd), **f32) for _ in range(3))
        d.body_pos_w = torch.zeros((N, nb, 3), **f32)
        d.body_quat_w = torch.zeros((N, nb, 4), **f32)
        d.body_quat_w[..., 0] = 1.0
        d.body_lin_vel_w = torch.zeros((N, nb, 3), **f32)
        d.body_ang_vel_w = torch.zeros((N, nb, 3), **f32)
        d.soft_joint_pos_limits = torch.tensor([-soft_limit, soft_limit], **f32).repeat(N, nd, 1)
        d.default_root_state = torch.zeros((N, 13), **f32)
        d.default_root_state[:, 2] = init_height
        d.default_root_state[:, 3] = 1.0
        d.default_joint_pos, d.default_joint_vel = torch.zeros((N, nd), **f32), torch.zeros((N, nd), **f32)
        self.data = d
        self._ALL_INDICES = torch.arange(N, dtype=torch.long, device=self.device)
        self._target = torch.zeros((N, nd), **f32)
        # fixed body offsets around the root so key bodies move with it
        self._offsets = torch.randn((nb, 3), generator=self._gen, **f32) * 0.3
        self._offsets[self._root] = 0.0
        self.write_root_link_pose_to_sim(d.default_root_state[:, :7], self._ALL_INDICES)

    # ---- Articulation surface used by the env ---------------------------------------------------------
    def reset(self, env_ids=None):
        ids = self._ALL_INDICES if env_ids is None else env_ids
        self.data.joint_acc[ids] = 0.0

    def reset_masked(self, mask: torch.Tensor):
        """``reset(env_ids)`` for a device-side bool mask (no id list, no sync)."""
        self.data.joint_acc.masked_fill_(mask[:, None], 0.0)

    def set_joint_position_target(self, target: torch.Tensor):
        self._target = target

    def write_root_link_pose_to_sim(self, pose: torch.Tensor, env_ids):
        d = self.data
        d.body_pos_w[env_ids] = pose[:, None, 0:3] + self._offsets[None]
        d.body_quat_w[env_ids] = pose[:, None, 3:7].expand(-1, d.body_quat_w.shape[1], -1)

    def write_root_com_velocity_to_sim(self, vel: torch.Tensor, env_ids):
        d = self.data
        d.body_lin_vel_w[env_ids] = vel[:, None, 0:3].expand(-1, d.body_lin_vel_w.shape[1], -1)
        d.body_ang_vel_w[env_ids] = vel[:, None, 3:6].expand(-1, d.body_ang_vel_w.shape[1], -1)

    def write_joint_state_to_sim(self, joint_pos, joint_vel, joint_ids, env_ids):
        self.data.joint_pos[env_ids] = joint_pos
        self.data.joint_vel[env_ids] = joint_vel

    def write_reset_compact(self, env_ids: torch.Tensor, count: torch.Tensor, root_state: torch.Tensor, joint_pos: torch.Tensor,
                            joint_vel: torch.Tensor):
        """Device-only counterpart of the three write_*_to_sim calls: rows i < count of the compact arrays go to env
        ``env_ids[i]``; nothing is read back to the host (rows >= count are routed to a scratch row)."""
        d, N = self.data, self.num_envs
        valid = torch.arange(env_ids.numel(), device=self.device) < count
        tgt = torch.where(valid, env_ids, torch.full_like(env_ids, N))  # N = scratch row
        def scatter(dst, src):
            pad = torch.cat([dst, dst[:1]], dim=0)
            pad.index_copy_(0, tgt, src)
            dst.copy_(pad[:N])
        nb = d.body_pos_w.shape[1]
        scatter(d.joint_pos, joint_pos)
        scatter(d.joint_vel, joint_vel)
        scatter(d.body_pos_w, root_state[:, None, 0:3] + self._offsets[None])
        scatter(d.body_quat_w, root_state[:, None, 3:7].expand(-1, nb, -1))
        scatter(d.body_lin_vel_w, root_state[:, None, 7:10].expand(-1, nb, -1))
        scatter(d.body_ang_vel_w, root_state[:, None, 10:13].expand(-1, nb, -1))
        mask = torch.zeros(N + 1, dtype=torch.bool, device=self.device)
        mask.index_fill_(0, tgt, True)  # (mask[tgt] = True stages its scalar through a blocking H2D copy)
        d.joint_acc.masked_fill_(mask[:N, None], 0.0)

    # ---- toy physics ------------------------------------------------------------------------------------
    def step(self):
        d, dt = self.data, self.dt
        d.joint_acc.copy_(400.0 * (self._target - d.joint_pos) - 40.0 * d.joint_vel)
        d.joint_vel.add_(d.joint_acc, alpha=dt)
        d.joint_pos.add_(d.joint_vel, alpha=dt)
        kick = torch.randn((self.num_envs, 1, 3), generator=self._gen, device=self.device) * 0.05
        d.body_lin_vel_w.add_(kick)
        d.body_lin_vel_w[..., 2] -= 9.81 * dt * 0.05
        d.body_pos_w.add_(d.body_lin_vel_w, alpha=dt)
        d.body_pos_w[..., 2].clamp_(min=0.0)
