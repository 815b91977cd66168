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
        self._in_step = False
        self._step_dev, self._graph = None, None  # device-side mirror of common_step_counter / captured step (capture_step)
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
                if self._graph is not None and int(seed) != self._reset_seed:
                    # a captured step has the old key baked into its launches: back to eager until capture_step() is called again.
                    # `graph_actions` stays the same static buffer, so `step(env.graph_actions)` keeps working (eagerly).
                    import warnings

                    warnings.warn("reset(seed=...) with a new seed dropped the captured step graph: the env steps eagerly until "
                                  "capture_step() is called again", RuntimeWarning, stacklevel=2)
                    self._graph = None
                self._reset_seed = int(seed)
        self._reset_idx(None)
        return self._get_observations(), self.extras

    def step(self, action: torch.Tensor):
        if self._graph is not None:
            return self._replay_step(action)
        return self._step_eager(action.to(self.device))

    def capture_step(self, warmup: int = 3):
        """Capture ``step()`` into a hipGraph; later ``step(actions)`` calls copy the actions into a static buffer and
        replay it.  Every launch of a device-reset step is asynchronous, allocation-free and free of host round trips, so
        the whole step -- hooks, state-provider writes and (synthetic) physics included -- is one graph; what changes from
        step to step lives on the device (the counter-based draws read a device-side step counter that the graph itself
        increments).  Needs ``device_reset=True``.  The returned observation / reward / done tensors are static buffers,
        overwritten by the next step (as ``extras["amp_obs"]`` always is).

        Side effects to know about: the ``warmup`` passes are REAL steps of this env on zero actions (episode lengths,
        resets, draw counters and physics advance by ``warmup`` steps; the recording pass itself executes nothing).  The
        graph bakes every pointer and the draw key of the moment: ``reset(seed=...)`` with a new seed drops the graph (the
        env steps eagerly until ``capture_step`` is called again), and tensors the hooks read must keep their addresses
        (the env's own buffers do; a state provider that re-allocates its arrays needs a new capture)."""
        if not getattr(self, "device_reset", False):
            raise ValueError("capture_step needs device_reset=True (the host-driven reset reads the reset count back)")
        n_act = int(self.cfg.action_space)
        self._graph_actions = torch.zeros((self.num_envs, n_act), dtype=torch.float32, device=self.device)
        # [0] the counter the step starts with, [1] the word a task's launches hand it over through (see _step_dev_in_launches)
        self._step_dev = torch.full((2,), self.common_step_counter, dtype=torch.int64, device=self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._step_eager(self._graph_actions)
        torch.cuda.current_stream(self.device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        gen = getattr(getattr(self, "robot", None), "_gen", None)
        if gen is not None and hasattr(g, "register_generator_state"):
            g.register_generator_state(gen)  # the synthetic articulation's torch.Generator draws inside the graph
        with torch.cuda.graph(g):
            out = self._step_eager(self._graph_actions)
        self.common_step_counter -= 1  # recording ran the Python bookkeeping of one step but no kernel
        self._graph = (g, out)
        return self

    def _replay_step(self, action: torch.Tensor):
        g, out = self._graph
        if action.data_ptr() != self._graph_actions.data_ptr():  # a policy may write straight into `graph_actions`: no copy then
            self._graph_actions.copy_(action)
        g.replay()
        self.common_step_counter += 1
        self._after_replay()
        return out[0], out[1], out[2], out[3], self.extras

    @property
    def graph_actions(self) -> torch.Tensor | None:
        """The captured step's static action buffer (``None`` before the first ``capture_step``): ``step(env.graph_actions)``
        after writing the actions into it replays without the per-step copy.  It outlives a dropped graph (``reset`` with a new
        seed): the same call then steps eagerly on it."""
        return getattr(self, "_graph_actions", None)

    def _after_replay(self):
        """Refresh per-step Python objects (``extras``) after a graph replay; tasks override."""

    def _step_eager(self, action: torch.Tensor):
        self._in_step = True   # hooks may defer work to a later hook of the same step (the reward-log means ride on the reset launch)
        self._pre_physics_step(action)
        for _ in range(self.cfg.decimation):
            self._apply_action()
            self.robot.step() if hasattr(self.robot, "step") else None
        if not getattr(self, "_pre_physics_counts_steps", False):  # (tasks on the engine: done by the pre-physics launch)
            self.episode_length_buf += 1
        self.common_step_counter += 1
        if self._step_dev is not None and not getattr(self, "_step_dev_in_launches", False):
            self._step_dev[:1] += 1   # (tasks on the engine advance it inside their pre-physics and reset launches: no launch of its own)
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
        self._in_step = False
        return self.obs_buf, self.reward_buf, self.reset_terminated, self.reset_time_outs, self.extras

    def close(self):
        pass
