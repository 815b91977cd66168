"""Minimal stand-in for ``isaaclab.envs.DirectRLEnv``: the vec-env template method the reference subclasses.

Isaac Lab is third-party and absent: this restates only the documented ``step`` ordering
the reference's hooks rely on [recalled, unpinned]:

    _pre_physics_step -> decimation x (_apply_action, physics) -> episode_length_buf += 1 -> _get_dones ->
    _get_rewards -> reset of (terminated | time_out) envs via _reset_idx -> _get_observations

``reset_buf.nonzero()`` (a host-synchronising ATen op in Isaac Lab) is replaced by the engine's ballot / prefix-sum
compaction; only the reset COUNT is read back, which is the same single sync ``len(reset_env_ids)`` costs there.
"""

from __future__ import annotations

import math

import torch


class DirectRLEnv:
    def __init__(self, cfg, render_mode: str | None = None, robot=None, **kwargs):
        self.cfg = cfg
        self.render_mode = render_mode
        self.num_envs = int(cfg.scene.num_envs)
        self.device = torch.device(getattr(cfg.sim, "device", "cuda:0"))
        self.physics_dt = float(cfg.sim.dt)
        self.step_dt = self.physics_dt * cfg.decimation
        self.max_episode_length_s = cfg.episode_length_s
        self.max_episode_length = math.ceil(cfg.episode_length_s / self.step_dt)
        self.episode_length_buf = torch.zeros(self.num_envs, dtype=torch.long, device=self.device)
        self.reset_terminated = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        self.reset_time_outs = torch.zeros_like(self.reset_terminated)
        self.reset_buf = torch.zeros_like(self.reset_terminated)
        self.extras = {}
        self.common_step_counter = 0
        self._given_robot = robot
        self.scene = type("Scene", (), {})()
        self.scene.env_origins = self._grid_origins(self.num_envs, float(cfg.scene.env_spacing)).to(self.device)
        self.sim = type("Sim", (), {"device": self.device})()
        self._setup_scene()

    @staticmethod
    def _grid_origins(n: int, spacing: float) -> torch.Tensor:
        side = math.ceil(math.sqrt(n))
        idx = torch.arange(n)
        xy = torch.stack([(idx // side).float(), (idx % side).float()], dim=1) * spacing
        xy -= xy.mean(dim=0, keepdim=True)
        return torch.cat([xy, torch.zeros(n, 1)], dim=1)

    # hooks a task implements ------------------------------------------------------------------------------
    def _setup_scene(self): ...
    def _pre_physics_step(self, actions): ...
    def _apply_action(self): ...
    def _get_observations(self): ...
    def _get_rewards(self): ...
    def _get_dones(self): ...

    def _reset_idx(self, env_ids):
        self.episode_length_buf[env_ids] = 0

    def _reset_on_device(self):
        raise NotImplementedError

    def _reset_buf(self) -> torch.Tensor:
        """``reset_terminated | reset_time_outs``; tasks on the engine return the mask their DONES launch wrote."""
        return self.reset_terminated | self.reset_time_outs

    def _compact_reset_ids(self) -> torch.Tensor:
        """Ascending int64 ids of ``reset_buf``; tasks on the engine override this with the fused tile counts."""
        from ..engine import reset_compact

        ids, count = reset_compact(self.reset_buf)
        return ids[: int(count)]

    # gym-style API ---------------------------------------------------------------------------------------
    def reset(self, seed: int | None = None, options=None):
        if seed is not None:
            import numpy as np

            np.random.seed(seed)
            torch.manual_seed(seed)
            if hasattr(self, "_reset_seed"):  # the engine's counter-based draws follow the run seed as torch.rand does
                self._reset_seed = int(seed)
        self._reset_idx(None)
        return self._get_observations(), self.extras

    def step(self, action: torch.Tensor):
        action = action.to(self.device)
        self._pre_physics_step(action)
        for _ in range(self.cfg.decimation):
            self._apply_action()
            self.robot.step() if hasattr(self.robot, "step") else None
        self.episode_length_buf += 1
        self.common_step_counter += 1
        self.reset_terminated, self.reset_time_outs = self._get_dones()
        self.reset_buf = self._reset_buf()
        self.reward_buf = self._get_rewards()
        if getattr(self, "device_reset", False):
            self._reset_on_device()   # engine path: no count read-back, no host RNG
        else:
            reset_env_ids = self._compact_reset_ids()
            if len(reset_env_ids) > 0:
                self._reset_idx(reset_env_ids)
        self.obs_buf = self._get_observations()
        return self.obs_buf, self.reward_buf, self.reset_terminated, self.reset_time_outs, self.extras

    def close(self):
        pass
