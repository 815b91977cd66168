"""G1AmpEnv / HumanoidAmpEnv: the reference's task envs with every hook routed to the HIP engine.

Same hook names, attributes and agent-facing surface as the reference (g1_amp_env.py:22-486,
humanoid_amp_env.py:22-248): ``step`` / ``reset``, ``extras["amp_obs"]`` (a VIEW of ``amp_observation_buffer``),
``amp_observation_space`` / ``amp_observation_size``, ``collect_reference_motions(num_samples, current_times,
motion_ids)``.  Each hook is one engine call:

    _get_dones        -> amp_env_step(DONES | REWARD)  (+ per-tile reset counts for the compaction): DirectRLEnv.step calls
    _get_rewards         _get_rewards right after _get_dones on the same state, so ONE launch serves both hooks
                         (_get_rewards on its own still launches REWARD)
    reset ids         -> amp_reset_compact_tiles (replaces reset_buf.nonzero())
    _reset_idx        -> amp_reset_reference_state + amp_collect_reference(scatter into the AMP buffer)
    _get_observations -> amp_env_step(OBS)
    collect_reference_motions -> amp_collect_reference

The humanoid env follows the G1 calling convention for ``sample_times`` / ``motion_ids`` (the reference's
humanoid env was not updated for this fork's tuple-returning ``sample_times`` and raises for K > 1).
"""

from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from .. import _native as nat
from ..engine import EnvStepConfig, EnvStepKernel, LazyRewardLog, REWARD_TERMS, command_step, reward_log_means
from ..motions import MotionLoader
from ..robots import G1_BODY_NAMES, G1_JOINT_NAMES, G1_KEY_BODY_NAMES, HUMANOID_KEY_BODY_NAMES
from .direct_rl_env import DirectRLEnv
from .sim import SyntheticArticulation


class Box(SimpleNamespace):
    """Shape-only stand-in for ``gym.spaces.Box`` (gymnasium is not a dependency of the hot path)."""

    def __init__(self, low, high, shape):
        super().__init__(low=low, high=high, shape=tuple(shape), dtype=np.float32)


class _AmpEnv(DirectRLEnv):
    KEY_BODY_NAMES: list = []
    ROOT_BODY = "pelvis"   # body whose state initialises the root on reset (g1_amp_env.py:398 / humanoid :194)
    Z_LIFT = 0.05
    IS_G1 = True

    def __init__(self, cfg, render_mode: str | None = None, robot=None, log_rewards: bool = True,
                 device_reset: bool = False, reset_seed: int | None = None, env_offset: int | None = None, **kwargs):
        """``device_reset=True`` keeps the whole reset on the device (ids + count from the compaction kernel feed
        ``amp_reset_apply``; clip / time come from the engine's counter-based RNG keyed by ``reset_seed``): no host
        sync and no numpy RNG inside ``step()`` (``tests/test_gpu_reset_device.py`` steps under
        ``torch.cuda.set_sync_debug_mode("error")``).  Needs a state provider with ``write_reset_compact`` (the
        synthetic articulation has one); the reference's exact host-RNG sequence is then not reproduced (distribution
        is).

        ``reset_seed``: key of every counter-based draw (velocity commands, device-side reset clip / time).  ``None``
        follows the run's seed the way the reference's ``torch.rand`` does: ``cfg.seed`` if the config has one, else
        ``torch.initial_seed()`` (what ``torch.manual_seed`` set); ``env.reset(seed=...)`` re-keys it.
        ``env_offset``: global id of this shard's env 0 -- every draw is keyed by the GLOBAL env id, so a sharded run
        draws what the unsharded run draws and two ranks never draw the same stream.  ``None``: this rank's block of
        ``distributed.shard_bounds`` when ``torch.distributed`` is initialised (every rank holding ``num_envs`` envs),
        else 0."""
        self._log_rewards = bool(log_rewards)
        self.device_reset, self._reset_out = bool(device_reset), None
        if reset_seed is None:
            reset_seed = getattr(cfg, "seed", None)
        self._reset_seed = int(torch.initial_seed() if reset_seed is None else reset_seed)
        if env_offset is None:
            import torch.distributed as dist

            env_offset = dist.get_rank() * int(cfg.scene.num_envs) if dist.is_available() and dist.is_initialized() else 0
        self.env_offset = int(env_offset)
        self._bound, self._reward_fresh, self._reset_args, self._tick_args = {}, False, None, None
        self._pending_means, self._pending_log, self._log_args = None, None, nat.AmpRewardLogArgs()
        super().__init__(cfg, render_mode, robot=robot, **kwargs)
        nat.require_gpu(self.device)
        data = self.robot.data
        lo, hi = data.soft_joint_pos_limits[0, :, 0], data.soft_joint_pos_limits[0, :, 1]
        self.action_offset = 0.5 * (hi + lo)
        self.action_scale = hi - lo

        self._motion_loader = MotionLoader(motion_file=self.cfg.motion_file, device=self.device)
        self.ref_body_index = data.body_names.index(self.cfg.reference_body)
        self.key_body_indexes = [data.body_names.index(n) for n in self.KEY_BODY_NAMES]
        self.motion_dof_indexes = self._motion_loader.get_dof_index(data.joint_names)
        self.motion_ref_body_index = self._motion_loader.get_body_index([self.cfg.reference_body])[0]
        self.motion_key_body_indexes = self._motion_loader.get_body_index(self.KEY_BODY_NAMES)
        frame = self._motion_loader.set_obs_layout(self.motion_dof_indexes, self.motion_ref_body_index,
                                                   self.motion_key_body_indexes)
        if frame != self.cfg.amp_observation_space:
            raise ValueError(f"cfg.amp_observation_space = {self.cfg.amp_observation_space}, but this robot / clip emits "
                             f"{frame}-float AMP frames (2*DoF + 1 + 6 + 3 + 3 + 12)")
        root_idx = self._motion_loader.get_body_index([self.ROOT_BODY])[0]
        if root_idx != self.motion_ref_body_index:
            raise ValueError("the reset root body must be the AMP reference body")

        K = self.cfg.num_amp_observations
        self.amp_observation_size = K * self.cfg.amp_observation_space
        self.amp_observation_space = Box(-np.inf, np.inf, (self.amp_observation_size,))
        self.key_body_obs_size = len(self.KEY_BODY_NAMES) * 3
        n_actor = getattr(self.cfg, "num_actor_observations", 1)
        kcfg = EnvStepConfig(
            n_dof=len(data.joint_names), num_amp_observations=K, max_episode_length=self.max_episode_length,
            n_key=len(self.KEY_BODY_NAMES), num_actor_observations=n_actor, use_last_actions=self.IS_G1,
            history_include_last_actions=getattr(self.cfg, "history_include_last_actions", True),
            history_include_command=getattr(self.cfg, "history_include_command", True),
            early_termination=self.cfg.early_termination, termination_height=self.cfg.termination_height,
            reward_mode=1 if self.IS_G1 else 0,
            **({k: float(getattr(self.cfg, k)) for k in ("rew_termination", "rew_action_l2", "rew_joint_pos_limits",
                                                         "rew_joint_acc_l2", "rew_joint_vel_l2", "rew_track_vel")}
               if self.IS_G1 else {}))
        self._kernel = EnvStepKernel(kcfg, self.num_envs, self.device, log_reward_terms=self.IS_G1 and self._log_rewards)
        if self._kernel.policy_obs_size != self.cfg.observation_space:
            raise ValueError(f"cfg.observation_space = {self.cfg.observation_space} but the policy observation has "
                             f"{self._kernel.policy_obs_size} entries")
        # reference attribute names
        self.amp_observation_buffer = self._kernel.amp_observation_buffer
        self.observation_space = Box(-np.inf, np.inf, (self.cfg.observation_space,))
        self.action_space = Box(-np.inf, np.inf, (self.cfg.action_space,))
        f32 = dict(dtype=torch.float32, device=self.device)
        self.actions = torch.zeros((self.num_envs, self.cfg.action_space), **f32)
        self._target = torch.zeros((self.num_envs, self.cfg.action_space), **f32)
        self.action_offset, self.action_scale = self.action_offset.contiguous(), self.action_scale.contiguous()
        self._pre_args = None
        self.last_actions = torch.zeros((self.num_envs, self.cfg.action_space), **f32)
        self.command_target_speed = torch.zeros((self.num_envs, 2), **f32)
        self.command_time_left = torch.zeros(self.num_envs, **f32)
        self.motion_ids = torch.zeros(self.num_envs, dtype=torch.long, device=self.device)
        self.motion_start_times = torch.zeros(self.num_envs, **f32)
        if n_actor > 1:
            self.actor_obs_hist_per_frame = self._kernel.actor_hist_per_frame
            self.actor_obs_history_buffer = self._kernel.actor_obs_history_buffer
            self._just_reset_mask = self._kernel.just_reset_mask

    # ---- scene --------------------------------------------------------------------------------------------
    def _make_robot(self):
        raise NotImplementedError

    def _setup_scene(self):
        self.robot = self._given_robot if self._given_robot is not None else self._make_robot()

    # ---- engine views of the simulator state --------------------------------------------------------------
    def _sim_views(self):
        d, r = self.robot.data, self.ref_body_index
        return dict(joint_pos=d.joint_pos, joint_vel=d.joint_vel, root_pos=d.body_pos_w[:, r], root_quat=d.body_quat_w[:, r],
                    root_lin_vel=d.body_lin_vel_w[:, r], root_ang_vel=d.body_ang_vel_w[:, r])

    # ---- DirectRLEnv hooks ----------------------------------------------------------------------------------
    def _launch(self, phases: int) -> None:
        """``amp_env_step(phases)`` on this step's simulator views.  The argument structs are built once per phase set
        and reused for as long as every tensor keeps its address (``self.actions`` / ``last_actions`` are persistent
        buffers for that reason): a hook then costs one call across the C ABI, no marshalling."""
        d = self.robot.data
        key = (d.joint_pos.data_ptr(), d.joint_vel.data_ptr(), d.joint_acc.data_ptr(), d.body_pos_w.data_ptr(),
               d.body_quat_w.data_ptr(), d.body_lin_vel_w.data_ptr(), d.body_ang_vel_w.data_ptr(),
               d.soft_joint_pos_limits.data_ptr(), self.actions.data_ptr(), self.last_actions.data_ptr(),
               self.command_target_speed.data_ptr(), self.episode_length_buf.data_ptr(), id(self._kernel._disc_layout))
        hit = self._bound.get(phases)
        if hit is None or hit[0] != key:
            views = dict(episode_length=self.episode_length_buf, **self._sim_views())
            if phases & nat.AMP_PHASE_REWARD and self.IS_G1:
                views.update(joint_acc=d.joint_acc, actions=self.actions, soft_limits=d.soft_joint_pos_limits,
                             command=self.command_target_speed)
            if phases & nat.AMP_PHASE_OBS:
                views.update(body_pos=d.body_pos_w, key_body_indexes=self.key_body_indexes)
                if self.IS_G1:
                    views.update(command=self.command_target_speed, last_actions=self.last_actions)
            hit = self._bound[phases] = (key, self._kernel.bind(phases, **views))
        hit[1]()

    def _track_log(self, log):
        agent = getattr(self, "_skrl_agent", None)
        if agent is not None:  # the reference records to TensorBoard every step when an agent is attached (:307-315)
            if torch.cuda.is_current_stream_capturing():
                return  # reading the means is a host sync: never while a step is being recorded (_after_replay tracks the replays)
            try:
                for k, v in log.items():
                    agent.track_data(f"Reward / {k}", v)
            except Exception:
                pass

    def _step_args(self, behind: bool = False):
        """(step, step_dev) of the counter-based draws: the host counter, or -- once ``capture_step`` made the step a
        hipGraph -- its device-side mirror (the graph increments it; a baked host value would repeat every replay)."""
        if self._step_dev is not None:
            # the word the launch reads: [0] in front of DirectRLEnv.step's `common_step_counter += 1` (the pre-physics tick),
            # [1] = [0] + 1 behind it (the reset), written by the pre-physics launch; the reset launch copies [1] back to [0]
            return 0, self._step_dev.data_ptr() + (8 if behind and getattr(self, "_step_dev_in_launches", False) else 0)
        return self.common_step_counter & (2**64 - 1), None

    def _after_replay(self):
        log = self.extras.get("log")
        self.extras = {"amp_obs": self.amp_observation_buffer.view(-1, self.amp_observation_size)}
        if isinstance(log, LazyRewardLog):  # a fresh lazy view of the graph's static means tensor
            log = self.extras["log"] = log.renew()
            self._track_log(log)

    def _pre_physics(self, actions: torch.Tensor, tick) -> None:
        """``amp_pre_physics_step``: the reference's ``self.actions = actions.clone()``, the joint targets
        ``action_offset + action_scale * actions`` of ``_apply_action``, ``self.last_actions = self.actions.clone()`` (G1) and
        the command timers' tick as ONE launch into persistent buffers (g1_amp_env.py:142-173)."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or tuple(actions.shape) != tuple(self.actions.shape):
            actions = actions.to(dtype=torch.float32).contiguous().view(self.actions.shape)
        p = self._pre_args
        if p is None:
            a = nat.AmpPrePhysicsArgs()
            a.actions, a.target = self.actions.data_ptr(), self._target.data_ptr()
            a.last_actions = self.last_actions.data_ptr() if self.IS_G1 else None
            a.offset, a.scale = self.action_offset.data_ptr(), self.action_scale.data_ptr()
            a.num_envs, a.n_actions = self.num_envs, int(self.actions.shape[1])
            # DirectRLEnv.step's `episode_length_buf += 1` rides on this launch (nothing reads the buffer before _get_dones)
            a.episode_length = self.episode_length_buf.data_ptr()
            self._pre_physics_counts_steps = True
            p = self._pre_args = (a, nat.load().amp_pre_physics_step)
        if p[0].episode_length != self.episode_length_buf.data_ptr():  # the buffer was re-assigned: follow it
            p[0].episode_length = self.episode_length_buf.data_ptr()
        # a captured step keeps its step counter on the device: with the one-launch device reset behind it, this launch and the
        # reset launch advance it themselves (AmpPrePhysicsArgs.step_in / step_out, AmpResetArgs.step_dev_out)
        fold = self._step_dev is not None and getattr(self, "device_reset", False)
        self._step_dev_in_launches = fold
        p[0].step_in = self._step_dev.data_ptr() if fold else None
        p[0].step_out = self._step_dev.data_ptr() + 8 if fold else None
        p[0].actions_in = actions.data_ptr()
        with torch.cuda.device(self.device):
            nat.check(p[1](C.byref(p[0]), C.byref(tick) if tick is not None else None, nat.stream_ptr()), "amp_pre_physics_step")

    def _pre_physics_step(self, actions: torch.Tensor):
        self._pre_physics(actions, None)

    def _apply_action(self):
        self.robot.set_joint_position_target(self._target)  # computed by _pre_physics_step's launch

    def _get_dones(self):
        # dones AND task reward: DirectRLEnv.step calls _get_rewards next, on the same state (g1_amp_env.py:246-330)
        self._launch(nat.AMP_PHASE_DONES | nat.AMP_PHASE_REWARD)
        self._reward_fresh = True
        return self._kernel.died, self._kernel.time_out

    def _reset_buf(self):
        return self._kernel.reset_mask  # died | time_out, written by the DONES phase

    def _compact_reset_ids(self) -> torch.Tensor:
        ids, count = self._kernel.compact_resets()
        return ids[: int(count)]  # one scalar read-back, like len(nonzero()) in the reference

    def _get_observations(self) -> dict:
        self._reward_fresh = False
        self._launch(nat.AMP_PHASE_OBS)
        self.extras = {**{k: v for k, v in self.extras.items() if k == "log"},
                       "amp_obs": self.amp_observation_buffer.view(-1, self.amp_observation_size)}
        return {"policy": self._kernel.policy_obs}

    def _reset_idx(self, env_ids: torch.Tensor | None):
        if env_ids is None or len(env_ids) == self.num_envs:
            env_ids = self.robot._ALL_INDICES
        self.robot.reset(env_ids)
        super()._reset_idx(env_ids)
        strategy = self.cfg.reset_strategy
        if strategy == "default":
            root_state, joint_pos, joint_vel = self._reset_strategy_default(env_ids)
        elif strategy.startswith("random"):
            root_state, joint_pos, joint_vel = self._reset_strategy_random(env_ids, "start" in strategy)
        else:
            raise ValueError(f"Unknown reset strategy: {strategy}")
        self.robot.write_root_link_pose_to_sim(root_state[:, :7], env_ids)
        self.robot.write_root_com_velocity_to_sim(root_state[:, 7:], env_ids)
        self.robot.write_joint_state_to_sim(joint_pos, joint_vel, None, env_ids)
        self._after_reset(env_ids)

    def _after_reset(self, env_ids):
        pass

    def _reset_on_device(self):
        """Everything ``DirectRLEnv.step`` does between ``_get_rewards`` and ``_get_observations`` as ONE engine launch
        (``amp_reset_compact_apply``: reset-id compaction, clip / time draw, reference root / DoF state, K expert frames
        into ``amp_observation_buffer``, episode-length / last-action / just-reset clears, command resample) + the state
        provider's ``write_reset_compact``.  Ids and count never leave the device."""
        strategy = self.cfg.reset_strategy
        default = strategy == "default"
        if not default and not strategy.startswith("random"):
            raise ValueError(f"Unknown reset strategy: {strategy}")
        k = self._kernel
        key = (self.episode_length_buf.data_ptr(), self.last_actions.data_ptr(), self.command_target_speed.data_ptr(),
               self.command_time_left.data_ptr(), self.amp_observation_buffer.data_ptr())
        if self._reset_args is None or self._reset_args[0] != key:
            N, nd = self.num_envs, len(self.robot.data.joint_names)
            f32 = dict(dtype=torch.float32, device=self.device)
            o = self._reset_out = dict(root_state=torch.zeros((N, 13), **f32), dof_pos=torch.zeros((N, nd), **f32),
                                       dof_vel=torch.zeros((N, nd), **f32),
                                       motion_ids=torch.zeros(N, dtype=torch.int64, device=self.device),
                                       motion_times=torch.zeros(N, dtype=torch.float64, device=self.device))
            c, a = k.compact_args(), nat.AmpResetArgs()
            a.env_ids, a.count, a.max_n = c.ids, c.count, N
            a.start, a.K = int("start" in self.cfg.reset_strategy), int(self.cfg.num_amp_observations)
            a.env_origins, a.z_lift = self.scene.env_origins.data_ptr(), float(self.Z_LIFT)
            a.root_state, a.dof_pos, a.dof_vel = (o[n].data_ptr() for n in ("root_state", "dof_pos", "dof_vel"))
            a.amp_obs_buffer = self.amp_observation_buffer.data_ptr()
            a.motion_ids, a.motion_times = o["motion_ids"].data_ptr(), o["motion_times"].data_ptr()
            a.env_motion_ids, a.env_motion_start_times = self.motion_ids.data_ptr(), self.motion_start_times.data_ptr()
            a.env_offset = self.env_offset
            if default:
                # reset_strategy "default" (g1_amp_env.py:338-339, 362-369): default root / joint state, AMP buffer and commands untouched
                d = self.robot.data
                a.mode = nat.AMP_RESET_DEFAULT
                self._reset_defaults = tuple(t.to(torch.float32).contiguous() for t in
                                             (d.default_root_state, d.default_joint_pos, d.default_joint_vel))  # kept alive here
                a.default_root_state, a.default_joint_pos, a.default_joint_vel = (t.data_ptr() for t in self._reset_defaults)
            a.episode_length = self.episode_length_buf.data_ptr()
            if self.IS_G1:
                a.last_actions, a.n_actions = self.last_actions.data_ptr(), int(self.last_actions.shape[1])
                if getattr(self.cfg, "num_actor_observations", 1) > 1:
                    a.just_reset = self._just_reset_mask.data_ptr()
            cmd = self._command_args() if self.IS_G1 and not default else None
            self._reset_args = (key, c, a, cmd, nat.load().amp_reset_compact_apply, self._motion_loader._need_handle())
        _, c, a, cmd, fn, handle = self._reset_args
        a.seed = self._reset_seed & (2**64 - 1)
        a.step, a.step_dev = self._step_args(behind=True)
        a.step_dev_out = self._step_dev.data_ptr() if getattr(self, "_step_dev_in_launches", False) and a.step_dev else None
        if cmd is not None:
            cmd.seed, cmd.step, cmd.step_dev = a.seed, a.step, a.step_dev
        lg, means = None, self._pending_means
        if means is not None:
            lg = self._log_args
            lg.reward_terms, lg.n_terms, lg.means = k.reward_terms.data_ptr(), int(k.reward_terms.shape[0]), means.data_ptr()
            self._pending_means = None
        with torch.cuda.device(self.device):
            nat.check(fn(handle, C.byref(c), C.byref(a), C.byref(cmd) if cmd is not None else None,
                         C.byref(lg) if lg is not None else None, nat.stream_ptr()), "amp_reset_compact_apply")
        o = self._reset_out
        self.robot.write_reset_compact(k.reset_ids, k.reset_count, o["root_state"], o["dof_pos"], o["dof_vel"])
        log, self._pending_log = self._pending_log, None
        if log is not None:  # the means this step's _get_rewards deferred are enqueued now: the agent may look
            self._track_log(log)

    def _reset_strategy_default(self, env_ids):
        d = self.robot.data
        root_state = d.default_root_state[env_ids].clone()
        root_state[:, :3] += self.scene.env_origins[env_ids]
        return root_state, d.default_joint_pos[env_ids].clone(), d.default_joint_vel[env_ids].clone()

    def _reset_strategy_random(self, env_ids: torch.Tensor, start: bool = False):
        n = env_ids.shape[0]
        motion_ids, times = self._motion_loader.sample_times(n, start=start)  # host numpy RNG, as the reference
        ids_d = torch.from_numpy(np.asarray(motion_ids, dtype=np.int64)).to(self.device)
        t_d = torch.from_numpy(np.asarray(times, dtype=np.float64)).to(self.device)
        self.motion_ids[env_ids] = ids_d
        self.motion_start_times[env_ids] = t_d.float()
        root_state, dof_pos, dof_vel = self._motion_loader.reset_reference_state(
            t_d, ids_d, env_ids=env_ids, env_origins=self.scene.env_origins, z_lift=self.Z_LIFT)
        # expert history straight into amp_observation_buffer[env_ids] (g1_amp_env.py:414-419)
        self._motion_loader.collect_reference(t_d, ids_d, self.cfg.num_amp_observations, out=self.amp_observation_buffer,
                                              dst_rows=env_ids)
        self._resample_commands(env_ids, on_reset=True)
        return root_state, dof_pos, dof_vel

    def _resample_commands(self, env_ids, on_reset: bool):
        pass

    # ---- agent-facing -----------------------------------------------------------------------------------------
    def collect_reference_motions(self, num_samples: int, current_times: np.ndarray | None = None,
                                  motion_ids: np.ndarray | None = None) -> torch.Tensor:
        """Expert AMP observations [num_samples, K*D] (g1_amp_env.py:445-486).  ``current_times`` / ``motion_ids``
        may be numpy arrays or device tensors."""
        if current_times is None:
            motion_ids, current_times = self._motion_loader.sample_times(num_samples)
        return self._motion_loader.collect_reference(current_times, motion_ids, self.cfg.num_amp_observations)


class G1AmpEnv(_AmpEnv):
    KEY_BODY_NAMES = G1_KEY_BODY_NAMES
    ROOT_BODY, Z_LIFT, IS_G1 = "pelvis", 0.05, True

    def _make_robot(self):
        return SyntheticArticulation(self.num_envs, G1_JOINT_NAMES, G1_BODY_NAMES, self.device, root_body="pelvis",
                                     dt=self.physics_dt)

    def _command_args(self) -> nat.AmpCommandArgs:
        """Prebuilt ``AmpCommandArgs`` of the reset-side resample (seed / step are filled in per step)."""
        a = nat.AmpCommandArgs()
        a.command, a.time_left = self.command_target_speed.data_ptr(), self.command_time_left.data_ptr()
        lo, hi = (float(x) for x in self.cfg.track_vel_range)
        t_lo, t_hi = (float(x) for x in self.cfg.command_resampling_time_range)
        a.step_dt, a.vel_lo, a.vel_span, a.t_lo, a.t_span = float(self.step_dt), lo, hi - lo, t_lo, t_hi - t_lo
        a.env_offset = self.env_offset
        return a

    def _command_step(self, mode: int, **which):
        command_step(self.command_target_speed, self.command_time_left, mode=mode, step_dt=self.step_dt,
                     vel_range=self.cfg.track_vel_range, time_range=self.cfg.command_resampling_time_range,
                     seed=self._reset_seed, step=self.common_step_counter, env_offset=self.env_offset, **which)

    def _pre_physics_step(self, actions: torch.Tensor):
        # actions / targets / last_actions + the timers' tick with the resample of the expired envs (g1_amp_env.py:146-173):
        # one launch, no nonzero() sync; prebuilt arguments
        t = self._tick_args
        if t is None or t[0] != (self.command_target_speed.data_ptr(), self.command_time_left.data_ptr()):
            t = self._tick_args = ((self.command_target_speed.data_ptr(), self.command_time_left.data_ptr()), self._command_args())
        a = t[1]
        a.seed = self._reset_seed & (2**64 - 1)
        a.step, a.step_dev = self._step_args()
        self._pre_physics(actions, a)

    def _resample_commands(self, env_ids, on_reset: bool):
        # reset-side resample (g1_amp_env.py:421-439) for an explicit id list (the host-driven reset path)
        if on_reset and len(env_ids) > 0:
            self._command_step(nat.AMP_COMMAND_RESET, env_ids=env_ids.to(torch.int64).contiguous())

    def _get_rewards(self) -> torch.Tensor:
        if not self._reward_fresh:  # called on its own: the REWARD phase alone (reads the done bits of the last DONES launch)
            self._launch(nat.AMP_PHASE_REWARD)
        self._reward_fresh = False
        if self._log_rewards:
            # the reference's 6-8 .mean().item() syncs (:291-305) become one reduction launch; the read-back happens
            # only when somebody looks at extras["log"] (LazyRewardLog), so step() itself never waits for the device
            drop = () if self.cfg.rew_track_vel > 0.0 else ("rew_track_vel", "error_track_vel")
            if self.device_reset and self._in_step:
                # inside step() the means ride on the reset launch that follows (amp_reset_compact_apply, one launch fewer)
                # -- so the agent's tracking waits for that launch too (_reset_on_device): until then the tensor is unwritten
                means = self._pending_means = torch.empty(len(REWARD_TERMS), dtype=torch.float32, device=self.device)
                log = self._pending_log = LazyRewardLog(REWARD_TERMS, means, drop)
                self.extras["log"] = log
            else:
                means = reward_log_means(self._kernel.reward_terms)
                log = LazyRewardLog(REWARD_TERMS, means, drop)
                self.extras["log"] = log
                self._track_log(log)
        return self._kernel.reward

    def _after_reset(self, env_ids):
        self.last_actions[env_ids] = 0.0
        if getattr(self.cfg, "num_actor_observations", 1) > 1:
            self._just_reset_mask[env_ids] = True


class HumanoidAmpEnv(_AmpEnv):
    KEY_BODY_NAMES = HUMANOID_KEY_BODY_NAMES
    ROOT_BODY, Z_LIFT, IS_G1 = "torso", 0.15, False

    def _make_robot(self):
        names = np.load(self.cfg.motion_file.split(",")[0]) if self.cfg.motion_file.endswith(".npz") else None
        if names is None:
            raise ValueError("HumanoidAmpEnv needs an explicit robot when motion_file is not a single .npz")
        # the humanoid_28 asset's robot-side order is not recorded in the reference: clip order (unpinned)
        return SyntheticArticulation(self.num_envs, names["dof_names"].tolist(), names["body_names"].tolist(), self.device,
                                     root_body="torso", dt=self.physics_dt, init_height=1.0)

    def _get_rewards(self) -> torch.Tensor:
        if not self._reward_fresh:
            self._launch(nat.AMP_PHASE_REWARD)
        self._reward_fresh = False
        return self._kernel.reward
