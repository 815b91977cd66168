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
        self._scatter = None
        # fixed body offsets around the root so key bodies move with it
        self._offsets = torch.randn((nb, 3), generator=self._gen, **f32) * 0.3
        self._offsets[self._root] = 0.0
        self.write_root_link_pose_to_sim(d.default_root_state[:, :7], self._ALL_INDICES)

    # ---- Articulation surface used by the env ---------------------------------------------------------
    def reset(self, env_ids=None):
        ids = self._ALL_INDICES if env_ids is None else env_ids
        self.data.joint_acc[ids] = 0.0

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
        """Device-only counterpart of the three write_*_to_sim calls (+ ``reset``'s joint_acc clear): rows i < count of
        the compact arrays go to env ``env_ids[i]``.  One ``amp_scatter_rows`` launch over the seven state arrays, bounded
        by the device-side count: nothing is read back and rows >= count are never touched."""
        key = (env_ids.data_ptr(), count.data_ptr(), root_state.data_ptr(), joint_pos.data_ptr(), joint_vel.data_ptr())
        if self._scatter is None or self._scatter[0] != key:
            from ..engine import RowScatter

            d, nb = self.data, self.data.body_pos_w.shape[1]
            ops = [dict(dst=d.joint_pos, src=joint_pos), dict(dst=d.joint_vel, src=joint_vel), dict(dst=d.joint_acc, fill=0.0),
                   dict(dst=d.body_pos_w, src=root_state[:, 0:3], repeat=nb, add=self._offsets),
                   dict(dst=d.body_quat_w, src=root_state[:, 3:7], repeat=nb),
                   dict(dst=d.body_lin_vel_w, src=root_state[:, 7:10], repeat=nb),
                   dict(dst=d.body_ang_vel_w, src=root_state[:, 10:13], repeat=nb)]
            self._scatter = (key, RowScatter(ops, env_ids, count))
        self._scatter[1]()

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
