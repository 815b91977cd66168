"""Host-side objects around the per-step kernels of ``libamp_engine.so``.

* :class:`EnvStepKernel`   -- ``amp_env_step``: dones / task reward / observations of one env shard
* :func:`reset_compact`    -- ``amp_reset_compact*``: ascending reset ids (replaces ``nonzero``)
* :class:`AmpDiscriminator`-- ``amp_disc_*``: scaler + discriminator MLP + style reward

Everything here is plumbing: argument checks, pointer marshalling, output allocation.  No arithmetic.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _native as nat


# ---------------------------------------------------------------------------------------------------
# env step
# ---------------------------------------------------------------------------------------------------


@dataclass
class EnvStepConfig:
    """Values of the reference config that the kernels need (g1_amp_env_cfg.py:22-141, humanoid_amp_env_cfg.py:23-78)."""

    n_dof: int
    num_amp_observations: int
    max_episode_length: int
    n_key: int = 4
    num_actor_observations: int = 1
    use_last_actions: bool = True          # G1: policy obs = base | last_actions [| command]; humanoid: False
    history_include_last_actions: bool = True
    history_include_command: bool = True
    early_termination: bool = True
    termination_height: float = 0.5
    reward_mode: int = 1                   # 1: G1 task reward, 0: constant 1 (humanoid)
    rew_termination: float = 0.0
    rew_action_l2: float = 0.0
    rew_joint_pos_limits: float = 0.0
    rew_joint_acc_l2: float = 0.0
    rew_joint_vel_l2: float = 0.0
    rew_track_vel: float = 0.0
    track_sigma: float = 0.5               # g1_amp_env.py:263
    track_floor: float = 4.0               # g1_amp_env.py:264

    @property
    def use_command(self) -> bool:
        return self.use_last_actions and self.rew_track_vel > 0.0

    @property
    def amp_frame_size(self) -> int:
        return 2 * self.n_dof + 13 + 3 * self.n_key

    def to_c(self) -> nat.AmpEnvCfg:
        c = nat.AmpEnvCfg()
        c.n_dof, c.n_key = self.n_dof, self.n_key
        c.num_amp_observations, c.num_actor_observations = self.num_amp_observations, self.num_actor_observations
        c.use_last_actions, c.use_command = int(self.use_last_actions), int(self.use_command)
        c.history_include_last_actions = int(self.history_include_last_actions)
        c.history_include_command = int(self.history_include_command)
        c.early_termination, c.reward_mode = int(self.early_termination), int(self.reward_mode)
        c.max_episode_length = int(self.max_episode_length)
        c.termination_height = self.termination_height
        c.rew_termination, c.rew_action_l2 = self.rew_termination, self.rew_action_l2
        c.rew_joint_pos_limits, c.rew_joint_acc_l2 = self.rew_joint_pos_limits, self.rew_joint_acc_l2
        c.rew_joint_vel_l2 = self.rew_joint_vel_l2
        c.rew_track_vel, c.track_sigma, c.track_floor = float(self.rew_track_vel), float(self.track_sigma), float(self.track_floor)
        return c


REWARD_TERMS = ("total_reward", "rew_track_vel", "error_track_vel", "pub_termination", "pub_action_l2",
                "pub_joint_pos_limits", "pub_joint_acc_l2", "pub_joint_vel_l2")


class EnvStepKernel:
    """Owns the output buffers of one env shard and launches ``amp_env_step`` on views of the sim state."""

    def __init__(self, cfg: EnvStepConfig, num_envs: int, device, log_reward_terms: bool = False):
        self.cfg = cfg
        self.device = nat.require_gpu(device)
        self.num_envs = int(num_envs)
        self._lib = nat.load()
        self._c = cfg.to_c()
        P = int(self._lib.amp_policy_obs_size(C.byref(self._c)))
        per = int(self._lib.amp_actor_history_frame_size(C.byref(self._c)))
        if P < 0 or per < 0:
            raise nat.AmpEngineError("invalid env configuration: " + self._lib.amp_last_error().decode())
        self.policy_obs_size, self.actor_hist_per_frame = P, per
        N, K, D, dev = self.num_envs, cfg.num_amp_observations, cfg.amp_frame_size, self.device
        self.amp_observation_buffer = torch.zeros((N, K, D), device=dev)
        self.policy_obs = torch.zeros((N, P), device=dev)
        self.reward = torch.zeros(N, device=dev)
        self.died = torch.zeros(N, dtype=torch.bool, device=dev)
        self.time_out = torch.zeros(N, dtype=torch.bool, device=dev)
        self.reset_mask = torch.zeros(N, dtype=torch.bool, device=dev)
        self.tile_envs = int(self._lib.amp_env_step_tile_envs(C.byref(self._c), N))
        self.reset_tile_counts = torch.zeros((N + self.tile_envs - 1) // self.tile_envs, dtype=torch.int32, device=dev)
        self.reward_terms = torch.zeros((len(REWARD_TERMS), N), device=dev) if log_reward_terms else None
        if cfg.num_actor_observations > 1:
            self.actor_obs_history_buffer = torch.zeros((N, cfg.num_actor_observations - 1, per), device=dev)
            self.just_reset_mask = torch.zeros(N, dtype=torch.bool, device=dev)
        else:
            self.actor_obs_history_buffer, self.just_reset_mask = None, None
        self.reset_ids = torch.zeros(N, dtype=torch.int64, device=dev)
        self.reset_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.disc_input, self._disc_layout = None, None

    def attach_discriminator(self, disc: "AmpDiscriminator") -> torch.Tensor:
        """Fuse ``disc``'s input scaler into the OBS phase: every OBS launch also writes the scaled, zero-padded
        discriminator input in the layout the discriminator's GEMMs consume -- ``disc_input`` float32 ``[N, padded]``
        or float16 plane blocks ``[N, padded / 32, 2, 32]`` (feed it to ``disc.style_reward_prescaled``) -- which saves the
        separate scaler pass over ``amp_obs``.  Call again after ``disc.set_scaler`` / ``set_weights``."""
        lay = disc.input_layout()
        if lay.padded_dim < self.cfg.num_amp_observations * self.cfg.amp_frame_size:
            raise nat.AmpEngineError("discriminator input is narrower than K * D")
        blocks = lay.format == nat.AMP_DISC_INPUT_F16_BLOCKS  # per row and 32-column k-block: [p0 x 32 | p1 x 32] halves
        shape = (self.num_envs, lay.padded_dim // 32, 2, 32) if blocks else (self.num_envs, lay.padded_dim)
        dtype = torch.float16 if blocks else torch.float32
        if self.disc_input is None or tuple(self.disc_input.shape) != shape or self.disc_input.dtype != dtype:
            self.disc_input = torch.zeros(shape, dtype=dtype, device=self.device)  # padding columns stay zero
        self._disc_layout = lay
        return self.disc_input

    def _buffers(self) -> nat.AmpEnvBuffers:
        b = nat.AmpEnvBuffers()
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        b.amp_obs_buffer, b.policy_obs = p(self.amp_observation_buffer), p(self.policy_obs)
        b.actor_history, b.just_reset = p(self.actor_obs_history_buffer), p(self.just_reset_mask)
        b.reward, b.reward_terms = p(self.reward), p(self.reward_terms)
        b.died, b.time_out, b.reset_mask = p(self.died), p(self.time_out), p(self.reset_mask)
        b.reset_tile_counts = p(self.reset_tile_counts)
        b.amp_obs_read_next = int(bool(getattr(self, "amp_obs_read_next", False)))  # a discriminator that takes the raw rows reads them next
        if self.disc_input is not None:
            lay = self._disc_layout
            b.disc_input, b.disc_input_stride = self.disc_input.data_ptr(), lay.padded_dim  # elements per row
            b.disc_input_format, b.disc_plane_scale = lay.format, lay.plane_scale
            b.scaler_mean, b.scaler_den, b.scaler_clip = lay.mean_dev, lay.den_dev, lay.clip
        return b

    def launch(self, phases: int, *, joint_pos=None, joint_vel=None, joint_acc=None, actions=None, root_pos=None,
               root_quat=None, root_lin_vel=None, root_ang_vel=None, body_pos=None, key_body_indexes: Sequence[int] = (),
               soft_limits=None, episode_length=None, command=None, last_actions=None, reference=None) -> None:
        """One ``amp_env_step`` launch.  ``reference=(motion_loader, times, motion_ids, out)`` makes it the
        horizontally fused launch ``amp_env_step_with_reference``: the expert-motion sample
        (``motion_loader.collect_reference(times, motion_ids, K, out=out)``) runs in the same kernel, on its own
        workgroups, bit-identical to the separate call.  State arguments are torch views straight off the simulator:
        ``[N, n_dof]`` rows (any env stride), ``[N, 3|4]`` root views (e.g. ``body_pos_w[:, ref]``),
        ``body_pos`` = ``[N, B, 3]`` with ``key_body_indexes`` into B, ``soft_limits`` ``[N, n_dof, 2]`` or ``[n_dof, 2]``."""
        s = self.sim_state(joint_pos=joint_pos, joint_vel=joint_vel, joint_acc=joint_acc, actions=actions, root_pos=root_pos,
                           root_quat=root_quat, root_lin_vel=root_lin_vel, root_ang_vel=root_ang_vel, body_pos=body_pos,
                           key_body_indexes=key_body_indexes, soft_limits=soft_limits, episode_length=episode_length,
                           command=command, last_actions=last_actions)
        cfg, N = self.cfg, self.num_envs
        b = self._buffers()
        with torch.cuda.device(self.device):
            if reference is None:
                nat.check(self._lib.amp_env_step(C.byref(self._c), C.byref(s), C.byref(b), N, int(phases), nat.stream_ptr()),
                          "amp_env_step")
            else:
                loader, times, ids, out = reference
                n, K = self.check_reference(times, ids, out)
                nat.check(self._lib.amp_env_step_with_reference(C.byref(self._c), C.byref(s), C.byref(b), N, int(phases),
                                                                loader._handle, nat.dptr(times), nat.dptr(ids), n, K, nat.dptr(out),
                                                                nat.stream_ptr()), "amp_env_step_with_reference")

    def bind(self, phases: int, **views) -> "BoundEnvStep":
        """``launch(phases, **views)`` with everything marshalled ONCE: the returned object's ``__call__`` is a single
        ``amp_env_step`` across the C ABI on prebuilt structs (what the env hooks use step after step; valid for as long as
        the view tensors keep their addresses -- the env re-binds when a pointer changes -- and until the next
        ``attach_discriminator``)."""
        return BoundEnvStep(self, int(phases), self.sim_state(**views), self._buffers(), views)

    def check_reference(self, times, ids, out):
        """Argument checks of the fused expert-motion sample; returns (n_samples, K)."""
        cfg = self.cfg
        n, K = int(times.shape[0]), cfg.num_amp_observations
        if times.dtype != torch.float64 or ids.dtype != torch.int64 or ids.shape[0] != n:
            raise nat.AmpEngineError("reference times / ids must be float64 / int64 device tensors of equal length")
        if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != n * K * cfg.amp_frame_size:
            raise nat.AmpEngineError(f"reference output must be a contiguous float32 [{n}, {K * cfg.amp_frame_size}] tensor")
        return n, K

    def sim_state(self, *, joint_pos=None, joint_vel=None, joint_acc=None, actions=None, root_pos=None, root_quat=None,
                  root_lin_vel=None, root_ang_vel=None, body_pos=None, key_body_indexes: Sequence[int] = (), soft_limits=None,
                  episode_length=None, command=None, last_actions=None) -> nat.AmpSimState:
        """The ``AmpSimState`` view struct of :meth:`launch`'s state arguments (checked); reusable for as long as the
        tensors live and keep their addresses."""
        cfg, N = self.cfg, self.num_envs
        s = nat.AmpSimState()

        def put(field, t, inner):
            if t is None:
                return
            if t.shape[0] != N:
                raise nat.AmpEngineError(f"{field} has {t.shape[0]} rows, expected {N}")
            ptr, stride = nat.strided_view(t, inner, field)
            setattr(s, field, ptr)
            setattr(s, field + "_stride", stride)

        put("joint_pos", joint_pos, cfg.n_dof)
        put("joint_vel", joint_vel, cfg.n_dof)
        put("joint_acc", joint_acc, cfg.n_dof)
        put("actions", actions, cfg.n_dof)
        put("root_pos", root_pos, 3)
        put("root_quat", root_quat, 4)
        put("root_lin_vel", root_lin_vel, 3)
        put("root_ang_vel", root_ang_vel, 3)
        if body_pos is not None:
            if body_pos.dim() != 3 or body_pos.shape[0] != N or body_pos.shape[2] != 3 or body_pos.stride(2) != 1 \
                    or body_pos.stride(1) != 3 or body_pos.dtype != torch.float32 or body_pos.device.type != "cuda":
                raise nat.AmpEngineError("body_pos must be a float32 [N, B, 3] HIP tensor with packed bodies")
            if len(key_body_indexes) != cfg.n_key or max(key_body_indexes) >= body_pos.shape[1] or min(key_body_indexes) < 0:
                raise nat.AmpEngineError(f"need {cfg.n_key} key body indexes inside [0, {body_pos.shape[1]})")
            s.body_pos, s.body_pos_stride = body_pos.data_ptr(), int(body_pos.stride(0))
            for i, k in enumerate(key_body_indexes):
                s.key_body[i] = int(k)
        if soft_limits is not None:
            if soft_limits.dtype != torch.float32 or soft_limits.device.type != "cuda" or soft_limits.shape[-2:] != (cfg.n_dof, 2) \
                    or soft_limits.stride(-1) != 1 or soft_limits.stride(-2) != 2:
                raise nat.AmpEngineError("soft_limits must be a float32 [N, n_dof, 2] or [n_dof, 2] HIP tensor")
            s.soft_limits = soft_limits.data_ptr()
            s.soft_limits_stride = int(soft_limits.stride(0)) if soft_limits.dim() == 3 else 0
            if soft_limits.dim() == 3 and soft_limits.shape[0] != N:
                raise nat.AmpEngineError("soft_limits has the wrong number of envs")
        s.episode_length = nat.dptr(episode_length, torch.int64, "episode_length").value if episode_length is not None else None
        s.command = nat.dptr(command, torch.float32, "command").value if command is not None else None
        s.last_actions = nat.dptr(last_actions, torch.float32, "last_actions").value if last_actions is not None else None
        for name, t, shape in (("episode_length", episode_length, (N,)), ("command", command, (N, 2)),
                               ("last_actions", last_actions, (N, cfg.n_dof))):
            if t is not None and tuple(t.shape) != shape:
                raise nat.AmpEngineError(f"{name} must have shape {shape}, got {tuple(t.shape)}")
        return s

    def compact_args(self) -> nat.AmpCompactArgs:
        """``AmpCompactArgs`` of this shard's reset-id compaction (for the fused tail / ``amp_hot_step``)."""
        c = nat.AmpCompactArgs()
        c.mask, c.tile_counts = self.reset_mask.data_ptr(), self.reset_tile_counts.data_ptr()
        c.tile_envs, c.num_envs = self.tile_envs, self.num_envs
        c.ids, c.count = self.reset_ids.data_ptr(), self.reset_count.data_ptr()
        return c

    def compact_resets(self):
        """Ascending reset ids from ``reset_mask`` using the tile counts of the last DONES launch.
        Returns (ids buffer [N] int64, count [1] int64), both on the device: no host sync here."""
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_reset_compact_tiles(nat.dptr(self.reset_mask), nat.dptr(self.reset_tile_counts),
                                                        self.tile_envs, self.num_envs, nat.dptr(self.reset_ids), nat.dptr(self.reset_count),
                                                        nat.stream_ptr()), "amp_reset_compact_tiles")
        return self.reset_ids, self.reset_count


class BoundEnvStep:
    """One ``amp_env_step`` launch on prebuilt arguments (:meth:`EnvStepKernel.bind`)."""

    __slots__ = ("_fn", "_args", "_keep", "_dev", "_index", "phases")

    def __init__(self, kernel: EnvStepKernel, phases: int, state, bufs, keep):
        self._fn = kernel._lib.amp_env_step
        self._args = (C.byref(kernel._c), C.byref(state), C.byref(bufs), kernel.num_envs, phases)
        self._keep = (kernel, state, bufs, keep)  # the structs are referenced by pointer; the tensors by address
        self._dev, self._index, self.phases = kernel.device, kernel.device.index or 0, phases

    def __call__(self) -> None:
        if torch.cuda.current_device() == self._index:
            rc = self._fn(*self._args, torch.cuda.current_stream().cuda_stream)
        else:
            with torch.cuda.device(self._dev):
                rc = self._fn(*self._args, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            nat.check(rc, "amp_env_step")


def reset_compact(mask: torch.Tensor):
    """Stand-alone ``mask.nonzero().squeeze(-1)``: returns (ids [N] int64 buffer, count [1] int64) on the device."""
    lib = nat.load()
    dev = nat.require_gpu(mask.device)
    if mask.dim() != 1 or mask.dtype not in (torch.bool, torch.uint8) or not mask.is_contiguous():
        raise nat.AmpEngineError("mask must be a contiguous 1-D bool / uint8 tensor")
    n = mask.numel()
    ids = torch.empty(n, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(int(lib.amp_reset_compact_workspace_bytes(n)), 4), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(lib.amp_reset_compact(nat.dptr(mask), n, nat.dptr(ids), nat.dptr(count), nat.dptr(ws), nat.stream_ptr()),
                  "amp_reset_compact")
    return ids, count


class RowScatter:
    """``amp_scatter_rows`` on prebuilt arguments: ``dst[ids[i]] = src[i]`` for ``i < count`` with ids and count on the
    device -- the ``tensor[env_ids] = rows`` of the reset path without a count read-back.  ``ops`` is a list of dicts
    ``dict(dst=[N, ...] float32, src=[max_n, width] float32 or None, fill=0.0, repeat=1, add=[repeat, width] or None)``;
    a ``dst`` row is ``repeat`` back-to-back copies of the ``width`` source floats (``+ add``)."""

    def __init__(self, ops, ids: torch.Tensor, count: torch.Tensor):
        self._lib = nat.load()
        dev = nat.require_gpu(ids.device)
        arr = (nat.AmpScatterRows * len(ops))()
        for a, o in zip(arr, ops):
            dst, src, add, repeat = o["dst"], o.get("src"), o.get("add"), int(o.get("repeat", 1))
            if dst.dtype != torch.float32 or dst.stride(-1) != 1 or (dst.dim() > 1 and not dst[0].is_contiguous()):
                raise nat.AmpEngineError("RowScatter: dst must be float32 with contiguous rows")
            per = int(dst[0].numel())
            if per % repeat:
                raise nat.AmpEngineError("RowScatter: dst row length is not a multiple of repeat")
            a.dst, a.dst_stride, a.width, a.repeat = dst.data_ptr(), int(dst.stride(0)), per // repeat, repeat
            if src is not None:
                if src.dtype != torch.float32 or src.dim() != 2 or src.shape[1] != a.width or src.stride(1) != 1:
                    raise nat.AmpEngineError(f"RowScatter: src must be float32 [max_n, {a.width}]")
                a.src, a.src_stride = src.data_ptr(), int(src.stride(0))
            a.fill = float(o.get("fill", 0.0))
            if add is not None:
                if add.dtype != torch.float32 or not add.is_contiguous() or add.numel() != per:
                    raise nat.AmpEngineError("RowScatter: add must be a contiguous float32 [repeat, width] tensor")
                a.add = add.data_ptr()
        self._args = (arr, len(ops), nat.dptr(ids, torch.int64, "ids"), nat.dptr(count, torch.int64, "count"), int(ids.numel()))
        self._keep, self._dev = (ops, ids, count), dev

    def __call__(self) -> None:
        with torch.cuda.device(self._dev):
            nat.check(self._lib.amp_scatter_rows(*self._args, nat.stream_ptr()), "amp_scatter_rows")


def command_step(command: torch.Tensor, time_left: torch.Tensor, *, mode: int, step_dt: float, vel_range, time_range,
                 seed: int, step: int, env_offset: int = 0, reset_mask: Optional[torch.Tensor] = None,
                 env_ids: Optional[torch.Tensor] = None, count: Optional[torch.Tensor] = None) -> None:
    """``amp_command_step``: the velocity-command timers of ``G1AmpEnv._pre_physics_step`` (``mode`` =
    ``AMP_COMMAND_TICK``, g1_amp_env.py:146-167) or the reset-side resample (``AMP_COMMAND_RESET`` for the envs of
    ``reset_mask`` or of ``env_ids[:count]``, g1_amp_env.py:421-439), in place, one launch, no host sync.  Draws are
    counter-based: (seed, step, env_offset + env)."""
    lib = nat.load()
    dev = nat.require_gpu(command.device)
    N = int(time_left.shape[0])
    if tuple(command.shape) != (N, 2):
        raise nat.AmpEngineError(f"command must be [N, 2] for time_left [N]; got {tuple(command.shape)} / {tuple(time_left.shape)}")
    a = nat.AmpCommandArgs()
    a.command, a.time_left = nat.dptr(command, torch.float32, "command").value, nat.dptr(time_left, torch.float32, "time_left").value
    lo, hi = float(vel_range[0]), float(vel_range[1])
    t_lo, t_hi = float(time_range[0]), float(time_range[1])
    # the reference multiplies torch.rand by the python float (hi - lo): the span is formed in fp64, then rounded once
    a.step_dt, a.vel_lo, a.vel_span, a.t_lo, a.t_span = float(step_dt), lo, hi - lo, t_lo, t_hi - t_lo
    a.seed, a.step, a.env_offset = int(seed) & (2**64 - 1), int(step) & (2**64 - 1), int(env_offset)
    if reset_mask is not None:
        if reset_mask.dtype not in (torch.bool, torch.uint8) or reset_mask.numel() != N:
            raise nat.AmpEngineError("reset_mask must be a bool / uint8 tensor with one entry per env")
        a.reset_mask = nat.dptr(reset_mask, None, "reset_mask").value
    if env_ids is not None:
        a.env_ids, a.n_ids = nat.dptr(env_ids, torch.int64, "env_ids").value, int(env_ids.numel())
        a.count = nat.dptr(count, torch.int64, "count").value
    with torch.cuda.device(dev):
        nat.check(lib.amp_command_step(C.byref(a), N, int(mode), nat.stream_ptr()), "amp_command_step")


class LazyRewardLog(dict):
    """``extras["log"]`` of the G1 env (g1_amp_env.py:291-305) without the per-step host sync: holds the DEVICE tensor
    of term means (``amp_reward_log_means``, enqueued with the step) and reads it back the first time any entry is
    looked at.  A plain ``dict`` of python floats from then on, which is what the reference hands to skrl."""

    def __init__(self, names, means_dev: torch.Tensor, drop=()):
        super().__init__()
        self._names, self._means, self._drop = tuple(names), means_dev, frozenset(drop)
        self._src = means_dev

    def renew(self) -> "LazyRewardLog":
        """A fresh, unread log over the same device tensor (a replayed hipGraph rewrites it in place every step)."""
        return LazyRewardLog(self._names, self._src, self._drop)

    def _fill(self):
        m, self._means = self._means, None
        if m is not None:
            for k, v in zip(self._names, m.tolist()):  # the one read-back, paid by whoever looks
                if k not in self._drop:
                    dict.__setitem__(self, k, v)

    @property
    def materialized(self) -> bool:
        return self._means is None

    def __getitem__(self, k): self._fill(); return dict.__getitem__(self, k)
    def __iter__(self): self._fill(); return dict.__iter__(self)
    def __len__(self): self._fill(); return dict.__len__(self)
    def __contains__(self, k): self._fill(); return dict.__contains__(self, k)
    def __repr__(self): self._fill(); return dict.__repr__(self)
    def __eq__(self, o):
        self._fill()
        if isinstance(o, LazyRewardLog):
            o._fill()
        return dict.__eq__(self, o)
    def keys(self): self._fill(); return dict.keys(self)
    def values(self): self._fill(); return dict.values(self)
    def items(self): self._fill(); return dict.items(self)
    def get(self, k, d=None): self._fill(); return dict.get(self, k, d)
    def copy(self): self._fill(); return dict(self)
    # mutators materialise first too, so that the log behaves like the plain dict the reference hands to skrl
    def __setitem__(self, k, v): self._fill(); dict.__setitem__(self, k, v)
    def __delitem__(self, k): self._fill(); dict.__delitem__(self, k)
    def pop(self, *a): self._fill(); return dict.pop(self, *a)
    def popitem(self): self._fill(); return dict.popitem(self)
    def setdefault(self, k, d=None): self._fill(); return dict.setdefault(self, k, d)
    def update(self, *a, **kw): self._fill(); dict.update(self, *a, **kw)
    def clear(self): self._fill(); dict.clear(self)
    def __reversed__(self): self._fill(); return dict.__reversed__(self)
    def __or__(self, o): self._fill(); return dict(self) | o
    def __ror__(self, o): self._fill(); return o | dict(self)
    def __ior__(self, o): self._fill(); dict.update(self, o); return self
    def __ne__(self, o): return not self.__eq__(o)
    __hash__ = None


def reward_log_means(reward_terms: torch.Tensor) -> torch.Tensor:
    """Means over the envs of ``reward_terms [T, N]`` -> device tensor ``[T]`` (``amp_reward_log_means``; no sync)."""
    lib = nat.load()
    dev = nat.require_gpu(reward_terms.device)
    T, N = reward_terms.shape
    out = torch.empty(T, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(lib.amp_reward_log_means(nat.dptr(reward_terms, torch.float32, "reward_terms"), int(T), int(N), nat.dptr(out),
                                           nat.stream_ptr()), "amp_reward_log_means")
    return out


# ---------------------------------------------------------------------------------------------------
# discriminator
# ---------------------------------------------------------------------------------------------------


class AmpDiscriminator:
    """Inference half of skrl's AMP agent: ``amp_state_preprocessor`` -> discriminator MLP -> style reward
    (shape agents/skrl_g1_walk_amp_cfg.yaml:31-39; scales :88-95).  Weights are copied into the engine at
    construction; ``load_state_dict``-style updates go through :meth:`set_weights` (in place: ``amp_disc_set_weights``
    keeps the handle, the scaler and attached kernels / trainers valid)."""

    def __init__(self, weights: Sequence, device, *, running_mean: Optional[torch.Tensor] = None,
                 running_variance: Optional[torch.Tensor] = None, epsilon: float = 1e-8, clip_threshold: float = 5.0,
                 discriminator_reward_scale: float = 2.0, task_reward_weight: float = 0.0, style_reward_weight: float = 1.0,
                 precision: str = "f16x3"):
        """``precision`` selects the GEMM engine; both deliver fp32-class accuracy (<= 1e-6 on O(1) logits):
        "f16x3" (default: fp32 operands as two fp16 planes, three fp16 MFMAs per k-step into one fp32 accumulator,
        csrc/disc_gemm_f16.hpp) or "f32" (fp32 MFMA on fp32 operands: exact fma chain, ~2.7x slower)."""
        self.device = nat.require_gpu(device)
        self._mode = {"f16x3": nat.AMP_DISC_F16X3, "f32": nat.AMP_DISC_FP32}[precision]
        self.precision = precision
        self._lib = nat.load()
        self._handle = None
        self.reward_scale = float(discriminator_reward_scale)
        self.task_reward_weight, self.style_reward_weight = float(task_reward_weight), float(style_reward_weight)
        self.epsilon, self.clip_threshold = float(epsilon), float(clip_threshold)
        self._ws = None
        self.set_weights(weights)
        if running_mean is not None:
            self.set_scaler(running_mean, running_variance)

    def set_weights(self, weights: Sequence) -> None:
        """weights = [(W1 [1024,in], b1), (W2 [512,1024], b2), (W3 [1,512], b3)] in torch.nn.Linear layout."""
        if len(weights) != 3:
            raise ValueError("the discriminator has exactly three Linear layers")
        flat = []
        for w, b in weights:
            flat.append(w.detach().to(device=self.device, dtype=torch.float32).contiguous())
            flat.append(b.detach().to(device=self.device, dtype=torch.float32).contiguous())
        w1, b1, w2, b2, w3, b3 = flat
        if w2.shape[1] != w1.shape[0] or w3.shape != (1, w2.shape[0]) or b1.numel() != w1.shape[0] or b2.numel() != w2.shape[0] \
                or b3.numel() != 1:
            raise ValueError("inconsistent discriminator layer shapes")
        d = nat.AmpDiscDesc()
        d.in_dim, d.h1, d.h2 = int(w1.shape[1]), int(w1.shape[0]), int(w2.shape[0])
        d.w1, d.b1, d.w2, d.b2, d.w3, d.b3 = (t.data_ptr() for t in flat)
        if self._handle is not None:
            # in place: the handle, its scaler and every device pointer it handed out (attach_discriminator layouts,
            # an attached trainer) stay valid; `flat` is read in stream order on the current stream
            if (d.in_dim, d.h1, d.h2) != (self.in_dim, self._h1, self._h2):
                raise ValueError(f"set_weights: layer shapes {(d.in_dim, d.h1, d.h2)} differ from the discriminator's "
                                 f"{(self.in_dim, self._h1, self._h2)}; build a new AmpDiscriminator")
            with torch.cuda.device(self.device):
                nat.check(self._lib.amp_disc_set_weights(self._handle, C.byref(d), nat.stream_ptr()), "amp_disc_set_weights")
            self._keepalive = flat  # until the next update: the copies above are asynchronous
            return
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_disc_create(C.byref(d), nat.stream_ptr(), C.byref(h)), "amp_disc_create")
        self._handle = h
        self.in_dim, self._h1, self._h2 = d.in_dim, d.h1, d.h2
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_disc_set_precision(h, self._mode, nat.stream_ptr()), "amp_disc_set_precision")

    def set_scaler(self, running_mean: torch.Tensor, running_variance: torch.Tensor) -> None:
        """RunningStandardScaler statistics (kept in float64 like skrl does)."""
        m = running_mean.detach().to(device=self.device, dtype=torch.float64).contiguous()
        v = running_variance.detach().to(device=self.device, dtype=torch.float64).contiguous()
        if m.numel() != self.in_dim or v.numel() != self.in_dim:
            raise ValueError(f"scaler statistics must have {self.in_dim} entries")
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_disc_set_scaler(self._handle, nat.dptr(m), nat.dptr(v), self.epsilon, self.clip_threshold,
                                                    nat.stream_ptr()), "amp_disc_set_scaler")
            torch.cuda.current_stream().synchronize()  # m / v are temporaries

    def _workspace(self, rows: int, slot: int = 0) -> torch.Tensor:
        need = int(self._lib.amp_disc_workspace_bytes(self._handle, rows))
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need:
            ws = self._ws[slot] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def style_reward(self, amp_obs: torch.Tensor, task_reward: Optional[torch.Tensor] = None, *, want_logits: bool = False,
                     inputs_consumed: Optional[torch.cuda.Event] = None, workspace_slot: int = 0,
                     compact: Optional["EnvStepKernel"] = None):
        """amp_obs [M, K*D] -> dict(style [M,1], combined [M,1] (if task_reward given), logits [M,1] (optional)).

        ``inputs_consumed`` (a torch.cuda.Event) is recorded on the current stream right after the last read of
        ``amp_obs`` / ``task_reward``; ``workspace_slot`` selects one of several private workspaces so that two calls
        may be in flight on the same stream queue (see workloads.HotPath, overlap=True).  ``compact`` (the
        :class:`EnvStepKernel` whose DONES phase ran this step): its reset-id compaction rides on the call's tail launch
        (``amp_disc_style_reward_compact``), as with :meth:`style_reward_prescaled`.

        Where the whole batch takes the one-launch two-layer kernel (``plan_info(M)["raw_input"]``) that kernel reads the fp32
        rows itself: no scaler pass, no scaled copy (8-B aligned rows of even stride; otherwise the scaler pass runs)."""
        if amp_obs.dim() != 2 or amp_obs.shape[1] != self.in_dim or amp_obs.dtype != torch.float32 or amp_obs.stride(1) != 1:
            raise nat.AmpEngineError(f"amp_obs must be float32 [M, {self.in_dim}] with a contiguous last dim")
        nat.require_gpu(amp_obs.device)
        M = amp_obs.shape[0]
        f32 = dict(dtype=torch.float32, device=self.device)
        style = torch.empty((M, 1), **f32)
        logits = torch.empty((M, 1), **f32) if want_logits else None
        combined, task = None, None
        if task_reward is not None:
            task = task_reward.reshape(-1).to(**f32).contiguous()
            if task.numel() != M:
                raise nat.AmpEngineError("task_reward must have one entry per row")
            combined = torch.empty((M, 1), **f32)
        ws = self._workspace(M, workspace_slot)
        ev = C.c_void_p(inputs_consumed.cuda_event) if inputs_consumed is not None else C.c_void_p(None)
        if compact is not None:
            if inputs_consumed is not None:
                raise nat.AmpEngineError("style_reward: inputs_consumed and compact are mutually exclusive")
            c = compact.compact_args()
            with torch.cuda.device(self.device):
                nat.check(self._lib.amp_disc_style_reward_compact(
                    self._handle, C.c_void_p(amp_obs.data_ptr()), M, int(amp_obs.stride(0)) if M > 1 else self.in_dim, self.reward_scale,
                    nat.dptr(task), self.task_reward_weight, self.style_reward_weight, nat.dptr(logits), nat.dptr(style), nat.dptr(combined),
                    nat.dptr(ws), C.byref(c), nat.stream_ptr()), "amp_disc_style_reward_compact")
        else:
            with torch.cuda.device(self.device):
                nat.check(self._lib.amp_disc_style_reward(
                    self._handle, C.c_void_p(amp_obs.data_ptr()), M, int(amp_obs.stride(0)) if M > 1 else self.in_dim, self.reward_scale,
                    nat.dptr(task), self.task_reward_weight, self.style_reward_weight, nat.dptr(logits), nat.dptr(style), nat.dptr(combined),
                    nat.dptr(ws), ev, nat.stream_ptr()), "amp_disc_style_reward")
        out = {"style": style}
        if combined is not None:
            out["combined"] = combined
        if logits is not None:
            out["logits"] = logits
        return out

    def input_layout(self) -> nat.AmpDiscInputLayout:
        """Layout of the scaled input the GEMMs consume (``format``: fp32 rows or fp16 plane blocks, ``padded_dim``, the
        handle's fp32 scaler vectors as device pointers, ``clip``, ``plane_scale``)."""
        lay = nat.AmpDiscInputLayout()
        nat.check(self._lib.amp_disc_input_layout(self._handle, C.byref(lay)), "amp_disc_input_layout")
        return lay

    PLAN_NAMES = {0: "register-staged 64x64 / 128x128", 1: "LDS-DMA 128x128 both layers", 2: "LDS-DMA 256x256 + 256x128",
                  3: "LDS-DMA 256x256 both layers", 4: "LDS-DMA 128x128 + 64x128", 16: "fp32-MFMA 128x128x16"}

    def set_plan(self, *, fused: Optional[bool] = None, fused_min_rows: Optional[int] = None) -> None:
        """Per-handle kernel-plan override (``amp_disc_set_plan``; ``None`` = automatic): ``fused`` allows / forbids the one-launch
        two-layer kernel, ``fused_min_rows`` moves its threshold.  Every plan scores a row the same bit for bit."""
        f = -1 if fused is None else int(bool(fused))
        nat.check(self._lib.amp_disc_set_plan(self._handle, f, -1 if fused_min_rows is None else int(fused_min_rows)), "amp_disc_set_plan")

    def plan_info(self, rows: int) -> dict:
        """Which kernels a style-reward call of ``rows`` rows launches on this handle and which environment switches are set
        (``amp_disc_plan_info``): ``fused_rows`` leading rows on the one-launch two-layer kernel, the rest in ``chunk_rows``-row
        (layer 1, layer 2) launch pairs on ``plan``."""
        p = nat.AmpDiscPlanInfo()
        nat.check(self._lib.amp_disc_plan_info(self._handle, int(rows), C.byref(p)), "amp_disc_plan_info")
        env = [name for bit, name in ((1, "AMP_DISC_FUSED"), (2, "AMP_DISC_FUSED_MIN_ROWS"), (4, "AMP_TRAIN_FORK"), (8, "AMP_TRAIN_BK32"), (16, "AMP_TRAIN_F16_BIG"))
               if p.env_overrides & bit]
        return {"precision": "f32" if p.precision == nat.AMP_DISC_FP32 else "f16x3", "plan": int(p.plan),
                "plan_name": self.PLAN_NAMES.get(int(p.plan), "?"), "fused_rows": int(p.fused_rows), "chunk_rows": int(p.chunk_rows),
                "fused_min_rows": int(p.fused_min_rows), "env_overrides": env, "cu_count": int(p.cu_count),
                "raw_input": bool(p.raw_input)}

    def style_reward_prescaled(self, scaled: torch.Tensor, task_reward: Optional[torch.Tensor] = None, *,
                               want_logits: bool = False, compact: Optional["EnvStepKernel"] = None):
        """Same as :meth:`style_reward` for an input already scaled, padded and laid out as :meth:`input_layout`
        says (``EnvStepKernel.attach_discriminator``): float32 ``[M, padded]`` or float16 plane blocks ``[M, padded / 32, 2, 32]``.
        ``compact`` (the :class:`EnvStepKernel` whose DONES phase ran this step): its reset-id compaction
        (``compact_resets()``) rides on the finalize launch (``amp_disc_style_reward_prescaled_compact``), one launch
        fewer; ``compact.reset_ids`` / ``reset_count`` are valid once this call's work completes."""
        lay = self.input_layout()
        blocks = lay.format == nat.AMP_DISC_INPUT_F16_BLOCKS
        want = (torch.float16, (lay.padded_dim // 32, 2, 32)) if blocks else (torch.float32, (lay.padded_dim,))
        if scaled.dtype != want[0] or tuple(scaled.shape[1:]) != want[1] or not scaled.is_contiguous():
            raise nat.AmpEngineError(f"scaled input must be a contiguous {want[0]} [M, {', '.join(map(str, want[1]))}] tensor "
                                     "(see input_layout)")
        nat.require_gpu(scaled.device)
        M = scaled.shape[0]
        f32 = dict(dtype=torch.float32, device=self.device)
        style = torch.empty((M, 1), **f32)
        logits = torch.empty((M, 1), **f32) if want_logits else None
        combined, task = None, None
        if task_reward is not None:
            task = task_reward.reshape(-1).to(**f32).contiguous()
            if task.numel() != M:
                raise nat.AmpEngineError("task_reward must have one entry per row")
            combined = torch.empty((M, 1), **f32)
        ws = self._workspace(M)
        with torch.cuda.device(self.device):
            if compact is not None:
                c = compact.compact_args()
                nat.check(self._lib.amp_disc_style_reward_prescaled_compact(
                    self._handle, nat.dptr(scaled), M, self.reward_scale, nat.dptr(task), self.task_reward_weight,
                    self.style_reward_weight, nat.dptr(logits), nat.dptr(style), nat.dptr(combined), nat.dptr(ws), C.byref(c),
                    nat.stream_ptr()), "amp_disc_style_reward_prescaled_compact")
            else:
                nat.check(self._lib.amp_disc_style_reward_prescaled(self._handle, nat.dptr(scaled), M, self.reward_scale, nat.dptr(task),
                                                                    self.task_reward_weight, self.style_reward_weight,
                                                                    nat.dptr(logits), nat.dptr(style), nat.dptr(combined),
                                                                    nat.dptr(ws), nat.stream_ptr()),
                          "amp_disc_style_reward_prescaled")
        out = {"style": style}
        if combined is not None:
            out["combined"] = combined
        if logits is not None:
            out["logits"] = logits
        return out

    def _destroy(self):
        h, self._handle = self._handle, None
        if h is not None:
            self._lib.amp_disc_destroy(h)

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass


class AmpDiscriminatorTrainer:
    """Discriminator training step of skrl's AMP agent on the engine: BCE on
    (policy U replay) vs motion batches + logit regularisation + gradient penalty + weight decay, Adam.  Updates the
    attached :class:`AmpDiscriminator` in place (weights and, with ``use_scaler``, its running scaler).
    Hyper-parameter names / defaults: agents/skrl_g1_walk_amp_cfg.yaml:70,87-95."""

    LOSS_TERMS = ("prediction", "grad_penalty", "logit_reg", "weight_decay")

    def __init__(self, disc: AmpDiscriminator, *, batch_size: int = 4096, learning_rate: float = 5e-5,
                 discriminator_loss_scale: float = 5.0, discriminator_logit_regularization_scale: float = 0.05,
                 discriminator_gradient_penalty_scale: float = 5.0, discriminator_weight_decay_scale: float = 1e-4,
                 betas=(0.9, 0.999), adam_epsilon: float = 1e-8, use_scaler: bool = True, update_scaler: bool = True,
                 running_mean: Optional[torch.Tensor] = None, running_variance: Optional[torch.Tensor] = None,
                 current_count: float = 1.0, apply_update: bool = True, gemm_precision: str = "f16x3",
                 defer_refresh: bool = False):
        """``gemm_precision``: "f32" runs every GEMM on the fp32 matrix pipe; "f16x3" (default) runs six of the backward's ten
        products -- dH1 = dH2 W2, gW2 = dH2^T H1, gW1 = dH1^T Xs of the prediction loss and a1 = a2 W2, gW2 += a2^T e1,
        da2 = e1 W2^T of the gradient penalty: 44 of the backward's 48 GFLOP at K D = 166 -- at fp32 accuracy on the fp16 matrix pipe:
        two fp16 planes per operand, three MFMAs per product (the inference path's engine), planes written once per step with one
        power-of-two scale per operand from a bound that costs no pass on the step's critical path (a-priori bounds, or abs-max
        passes on the side stream).  The forward GEMMs stay on the fp32 pipe in both modes (they decide the ReLU masks).  Same
        gradients to 1e-5 of each tensor's scale; 0.66 -> 0.56 ms per step at K D = 166, 1.14 -> 0.89 at 830 (where the chain's other
        three products run there too)
        (profiles/r05_train_step.md).  Needs the forked step (gradient penalty on, hidden widths multiples of 128); otherwise
        the fp32 products run."""
        if gemm_precision not in ("f16x3", "f32"):
            raise ValueError(f"gemm_precision must be 'f16x3' or 'f32', got {gemm_precision!r}")
        self.disc, self.device, self._lib = disc, disc.device, nat.load()
        self.batch_size = int(batch_size)
        self.gemm_precision = gemm_precision
        c = nat.AmpDiscTrainCfg()
        c.max_rows_per_group = self.batch_size
        c.learning_rate, c.beta1, c.beta2, c.adam_epsilon = learning_rate, betas[0], betas[1], adam_epsilon
        c.loss_scale, c.logit_reg_scale = discriminator_loss_scale, discriminator_logit_regularization_scale
        c.grad_penalty_scale, c.weight_decay_scale = discriminator_gradient_penalty_scale, discriminator_weight_decay_scale
        c.scaler_epsilon, c.scaler_clip = disc.epsilon, disc.clip_threshold
        c.use_scaler, c.update_scaler, c.apply_update = int(use_scaler), int(update_scaler), int(apply_update)
        c.gemm_f16x3 = int(gemm_precision == "f16x3")
        # defer_refresh: the steps skip refreshing what the discriminator derives from its weights / scaler for INFERENCE (fp16
        # planes, plane scales, fp32 scaler vectors: ~45 us of small launches per step); call refresh() before the next style reward
        c.defer_refresh = int(bool(defer_refresh))
        self.defer_refresh = bool(defer_refresh)
        self.loss_scale = float(discriminator_loss_scale)
        m = None if running_mean is None else running_mean.detach().to(device=self.device, dtype=torch.float64).contiguous()
        v = None if running_variance is None else running_variance.detach().to(device=self.device, dtype=torch.float64).contiguous()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_disc_trainer_create(disc._handle, C.byref(c), nat.dptr(m), nat.dptr(v), float(current_count),
                                                        nat.stream_ptr(), C.byref(h)), "amp_disc_trainer_create")
            torch.cuda.current_stream().synchronize()
        self._handle = h
        self._n_params = sum(int(w.numel() + b.numel()) for w, b in self.weights())

    def weights(self):
        """Current [(W1, b1), (W2, b2), (W3, b3)] of the attached discriminator (device copies, Linear layout)."""
        d = self.disc
        f32 = dict(dtype=torch.float32, device=self.device)
        w1 = torch.empty((d._h1, d.in_dim), **f32); b1 = torch.empty(d._h1, **f32)
        w2 = torch.empty((d._h2, d._h1), **f32); b2 = torch.empty(d._h2, **f32)
        w3 = torch.empty((1, d._h2), **f32); b3 = torch.empty(1, **f32)
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_disc_get_weights(d._handle, *[nat.dptr(t) for t in (w1, b1, w2, b2, w3, b3)], nat.stream_ptr()),
                      "amp_disc_get_weights")
        return [(w1, b1), (w2, b2), (w3, b3)]

    def step(self, policy_states: torch.Tensor, replay_states: torch.Tensor, motion_states: torch.Tensor, *,
             want_grads: bool = False):
        """One update on three [B, K*D] batches of raw AMP observations.  Returns {"loss": total, terms..., "grads"?}
        as device tensors (no host sync)."""
        B = policy_states.shape[0]
        for name, t in (("policy", policy_states), ("replay", replay_states), ("motion", motion_states)):
            if t.dim() != 2 or t.shape != (B, self.disc.in_dim) or t.dtype != torch.float32 or t.stride(1) != 1 \
                    or t.stride(0) != policy_states.stride(0):
                raise nat.AmpEngineError(f"{name}_states must be float32 [B, {self.disc.in_dim}] with the same row stride")
            nat.require_gpu(t.device)
        loss = torch.empty(5, dtype=torch.float32, device=self.device)  # four terms + the scaled total
        grads = torch.empty(self._n_params, dtype=torch.float32, device=self.device) if want_grads else None
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_disc_train_step(self._handle, C.c_void_p(policy_states.data_ptr()),
                                                    C.c_void_p(replay_states.data_ptr()), C.c_void_p(motion_states.data_ptr()), B,
                                                    int(policy_states.stride(0)), nat.dptr(loss), nat.dptr(grads), nat.stream_ptr()),
                      "amp_disc_train_step")
        out = dict(zip(self.LOSS_TERMS, loss.unbind(0)))
        out["loss"] = loss[4]  # loss_scale * (sum of the four terms), formed on the device by the step's last kernel
        if want_grads:
            out["grads"] = grads
        return out

    def refresh(self) -> None:
        """With ``defer_refresh=True``: bring the attached discriminator's inference-side data (fp16 weight planes, plane scales,
        scaler vectors) up to date with the trained weights -- once per agent update instead of once per training step."""
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_disc_trainer_refresh(self._handle, nat.stream_ptr()), "amp_disc_trainer_refresh")

    def capture(self, want_grads: bool = False):
        """Capture the training step into a hipGraph over static input buffers (``self.static_inputs`` = policy, replay,
        motion ``[batch_size, K*D]``): :meth:`step_captured` copies three batches in and replays it.  Every launch of the
        step is asynchronous and allocation-free, and what advances from step to step (scaler sample count, Adam step and
        bias corrections) lives on the device, so a replay IS the next step.  The ~45 small launches of a step then cost
        the queue a graph's node-to-node latency instead of 45 host launches (profiles/NOTEBOOK_r01_r04.md section 7b).  Recording executes nothing:
        the trainer's state is untouched by :meth:`capture` itself."""
        B, dim = self.batch_size, self.disc.in_dim
        f32 = dict(dtype=torch.float32, device=self.device)
        self.static_inputs = tuple(torch.zeros((B, dim), **f32) for _ in range(3))
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self.step(*self.static_inputs, want_grads=want_grads)
        self._graph = (g, out)
        return self

    def step_captured(self, policy_states: torch.Tensor, replay_states: torch.Tensor, motion_states: torch.Tensor):
        """One update through the captured graph: returns the static output dict of :meth:`capture` (overwritten by the
        next replay)."""
        g, out = self._graph
        for dst, src in zip(self.static_inputs, (policy_states, replay_states, motion_states)):
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        g.replay()
        return out

    def scaler_state(self):
        """(running_mean, running_variance, current_count): fp64 device copies of the trainer's statistics."""
        n = self.disc.in_dim
        mean = torch.empty(n, dtype=torch.float64, device=self.device)
        var = torch.empty(n, dtype=torch.float64, device=self.device)
        cnt = C.c_double()
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_disc_trainer_scaler(self._handle, nat.dptr(mean), nat.dptr(var), C.byref(cnt), nat.stream_ptr()),
                      "amp_disc_trainer_scaler")
        return mean, var, float(cnt.value)

    def adam_state(self):
        """(exp_avg, exp_avg_sq, step): Adam's moments of (W1, b1, W2, b2, W3, b3) concatenated in their logical shapes (the
        layout of ``step(want_grads=True)["grads"]``) and the optimizer steps taken."""
        m = torch.empty(self._n_params, dtype=torch.float32, device=self.device)
        v = torch.empty(self._n_params, dtype=torch.float32, device=self.device)
        n = C.c_int64()
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_disc_trainer_adam_state(self._handle, nat.dptr(m), nat.dptr(v), C.byref(n), nat.stream_ptr()),
                      "amp_disc_trainer_adam_state")
        return m, v, int(n.value)

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h is not None:
            try:
                self._lib.amp_disc_trainer_destroy(h)
            except Exception:
                pass


# ---------------------------------------------------------------------------------------------------
# agent-side row stores (skrl RandomMemory: reply_buffer / motion_dataset)
# ---------------------------------------------------------------------------------------------------


def take_permuted_rows(rows: torch.Tensor, seed: int, epoch: int, first: int, count: int, *, out: Optional[torch.Tensor] = None,
                       return_indices: bool = False):
    """``out[i] = rows[pi(first + i)]`` for ``i < count`` with ``pi`` a pseudo-random permutation of the rows keyed by ``(seed, epoch)``
    and evaluated point-wise on the device (``amp_rows_take_permuted``: Feistel network + cycle walking; no sort, no index array):
    the epoch shuffle of an agent update's rollout rows.  Positions ``[m * per, m * per + batch)`` of one epoch are ``batch`` distinct
    rows, disjoint from every other minibatch's."""
    if rows.dim() != 2 or rows.dtype != torch.float32 or rows.stride(1) != 1:
        raise nat.AmpEngineError("rows must be a float32 [n, dim] tensor with a contiguous last dim")
    dev = nat.require_gpu(rows.device)
    if out is None:
        out = torch.empty((count, rows.shape[1]), device=dev)
    elif out.dim() != 2 or out.dtype != torch.float32 or tuple(out.shape) != (count, rows.shape[1]) or out.stride(1) != 1 or out.device != rows.device:
        raise nat.AmpEngineError("out must be a float32 [count, dim] tensor on the rows' device with a contiguous last dim")
    idx = torch.empty(count, dtype=torch.int64, device=dev) if return_indices else None
    with torch.cuda.device(dev):
        nat.check(nat.load().amp_rows_take_permuted(C.c_void_p(rows.data_ptr()), rows.shape[0], int(rows.stride(0)), rows.shape[1], int(seed),
                                                    int(epoch), int(first), int(count), C.c_void_p(out.data_ptr()), int(out.stride(0)),
                                                    nat.dptr(idx), nat.stream_ptr()), "amp_rows_take_permuted")
    return (out, idx) if return_indices else out


class AmpReplayBuffer:
    """Device ring buffer of AMP observation rows with skrl ``RandomMemory`` semantics [third-party, absent: parity
    unpinned]: :meth:`add_samples` writes a batch at the write head and wraps around, :meth:`sample` draws
    ``batch_size`` rows uniformly with replacement from the rows written so far.  Used for skrl AMP's ``reply_buffer``
    (``amp_replay_buffer_size`` 1 M) and ``motion_dataset`` (``amp_motion_dataset_size`` 200 k,
    agents/skrl_g1_walk_amp_cfg.yaml:44-58).  Draws are counter-based (Philox keyed by ``seed`` and a per-call draw
    counter): reproducible and free of host RNG; the row indices are available for inspection."""

    def __init__(self, memory_size: int, row_dim: int, device, seed: int = 0):
        self.device = nat.require_gpu(device)
        self._lib = nat.load()
        self.memory_size, self.row_dim, self.seed = int(memory_size), int(row_dim), int(seed)
        self._draw = 0
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_ring_create(self.memory_size, self.row_dim, C.byref(h)), "amp_ring_create")
        self._handle = h

    def __len__(self) -> int:
        return int(self._lib.amp_ring_size(self._handle))

    @property
    def memory_index(self) -> int:
        """Next write position (skrl's attribute name)."""
        return int(self._lib.amp_ring_head(self._handle))

    def add_samples(self, states: torch.Tensor) -> None:
        rows = states if states.dim() == 2 else states.reshape(-1, states.shape[-1])
        ptr, stride = nat.strided_view(rows, self.row_dim, "states")  # any row stride: views are appended without a copy
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_ring_append(self._handle, ptr, rows.shape[0], stride, nat.stream_ptr()), "amp_ring_append")

    def sample(self, batch_size: int, *, out: Optional[torch.Tensor] = None, return_indices: bool = False, first_row: int = 0):
        """``[batch_size, row_dim]`` rows (into ``out`` if given); each call advances the draw counter.  ``first_row``: where
        these rows sit in the minibatch they belong to (row i takes the variate of counter ``(first_row + i, draw)``): rank w of
        a multi-rank update draws ``batch / world`` rows at ``first_row = w * batch / world``."""
        if out is None:
            out = torch.empty((batch_size, self.row_dim), device=self.device)
        idx = torch.empty(batch_size, dtype=torch.int64, device=self.device) if return_indices else None
        with torch.cuda.device(self.device):
            nat.check(self._lib.amp_ring_sample(self._handle, self.seed, self._draw, int(first_row), batch_size, nat.dptr(out),
                                                int(out.stride(0)), nat.dptr(idx), nat.stream_ptr()), "amp_ring_sample")
        self._draw += 1
        return (out, idx) if return_indices else out

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h is not None and getattr(self, "_lib", None) is not None:
            self._lib.amp_ring_destroy(h)


class AmpDiscriminatorUpdate:
    """The discriminator half of skrl ``AMP._update`` as a device-side data flow [skrl is third-party and absent:
    restated, parity unpinned; hyper-parameters agents/skrl_g1_walk_amp_cfg.yaml:64-66,91-95]:

    * per learning epoch the rollout's AMP states are shuffled and split into ``mini_batches`` minibatches
      (``memory.sample_all``); the first ``batch_size`` rows of a minibatch are the policy batch;
    * each minibatch trains against ``batch_size`` rows drawn from the replay buffer (the policy batch itself while the
      buffer is empty) and ``batch_size`` rows drawn from the motion dataset (:class:`AmpReplayBuffer` rings);
    * after the update the rollout's rows are appended to the replay buffer.

    Every draw is on the device and counter-based -- the ring draws (Philox) and the epoch shuffle (a point-wise pseudo-random
    permutation keyed by ``(seed, epoch)``: :func:`take_permuted_rows`; no host RNG, no torch generator) --; nothing is read back.

    **Multi-rank** (``group`` = a ``torch.distributed`` process group; the reference's ``--distributed`` mode,
    train.py:54-58,183-196, keeps one agent replica per GPU in step through skrl's gradient all-reduce): the trainer's
    ``batch_size`` is the GLOBAL minibatch.  Every rank contributes ``batch_size / world`` rows to each group of every training
    step -- policy rows from its own shuffled rollout, replay / motion rows from its own rings, drawn at
    ``first_row = rank * batch_size / world`` so that a minibatch consumes the same variates whatever the world size -- the
    contributions of the whole update travel in ONE all-gather (``distributed.UpdateExchange``), and every replica then
    takes the same ``learning_epochs * mini_batches`` steps on the same global minibatches: weights, Adam state and scaler
    stay bit-identical across ranks with no gradient traffic.  Compute per rank is that of a single-GPU update
    (``batch_size`` = ``discriminator_batch_size``); a trainer built with ``batch_size = world * discriminator_batch_size``
    instead reproduces the mean-of-rank-means gradient of skrl's all-reduce at ``world`` times the compute.  A group of one
    rank is bit-identical to ``group=None``."""

    GROUPS = ("policy", "replay", "motion")

    def __init__(self, trainer: AmpDiscriminatorTrainer, replay: AmpReplayBuffer, motion_dataset: AmpReplayBuffer, *,
                 learning_epochs: int = 6, mini_batches: int = 2, seed: int = 0, record_batches: bool = False, prefetch: bool = True,
                 group=None, take_rows=None):
        """``prefetch`` (default; single-rank flow only): the three batches of training step k + 1 (shuffle, row gather, two
        ring draws: ~45 us of sort / gather launches) are produced on a side stream while step k trains, into two alternating
        sets of static buffers; the same draws in the same order, so the rows are identical to the in-line flow
        (``prefetch=False``).  With a ``group`` every batch of the update is produced up front (the exchange needs them all)."""
        self.trainer, self.replay, self.motion_dataset = trainer, replay, motion_dataset
        self.learning_epochs, self.mini_batches = int(learning_epochs), int(mini_batches)
        # the epoch shuffle: take_rows(rows, seed, epoch, first, count, out=...) -- the engine's point-wise permutation (a test double
        # on CPU tensors passes its own restatement); `_epoch` counts the epochs of all updates so far
        self.seed, self._epoch = int(seed), 0
        self._take_rows = take_rows or take_permuted_rows
        self.record_batches = bool(record_batches)
        self.batches = []  # (policy, replay, motion) of every trainer step of the last update, when recording
        self.prefetch = bool(prefetch)
        self._side, self._bufs = None, None
        self.group, self.exchange = group, None
        self.time_phases, self._marks = False, None   # set time_phases: HIP events at the phase boundaries of every update
        if group is not None:
            import torch.distributed as dist

            world = dist.get_world_size(group)
            if trainer.batch_size % world != 0:
                raise nat.AmpEngineError(f"the global minibatch ({trainer.batch_size} rows) must divide over the {world} ranks of the group")

    def _mark(self, name=None):
        if self.time_phases:
            if name is None:
                self._marks = []
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append((name, ev))

    def phase_ms(self) -> dict:
        """With ``time_phases``: milliseconds of the last update's phases on the current stream (blocks until the update is done):
        ``produce`` (this rank's rows of every step), ``exchange`` (the all-gather + re-blocking copy, as far as it is not hidden
        under the replay append), ``train`` (the optimizer steps + refresh), ``total``."""
        if not self._marks:
            return {}
        self._marks[-1][1].synchronize()
        out = {name: self._marks[i][1].elapsed_time(ev) for i, (name, ev) in enumerate(self._marks[1:])}
        out["total"] = self._marks[0][1].elapsed_time(self._marks[-1][1])
        return out

    def update(self, rollout_amp_states: torch.Tensor):
        rows = rollout_amp_states.reshape(-1, rollout_amp_states.shape[-1])
        bs = self.trainer.batch_size
        self._mark()
        if self.group is not None:
            return self._update_exchanged(rows, bs)
        if rows.shape[0] < bs * self.mini_batches:
            raise nat.AmpEngineError(f"the rollout has {rows.shape[0]} rows; {self.mini_batches} minibatches of {bs} are needed")
        if len(self.motion_dataset) == 0:
            raise nat.AmpEngineError("the motion dataset is empty: fill it with collect_reference rows first")
        losses, self.batches = [], []
        per = rows.shape[0] // self.mini_batches
        if self.prefetch and self.learning_epochs * self.mini_batches > 0:
            return self._update_prefetched(rows, per, bs)
        for _ in range(self.learning_epochs):
            epoch, self._epoch = self._epoch, self._epoch + 1
            for mb in range(self.mini_batches):
                policy = self._take_rows(rows, self.seed, epoch, mb * per, bs)
                replay = self.replay.sample(bs) if len(self.replay) > 0 else policy
                motion = self.motion_dataset.sample(bs)
                losses.append(self.trainer.step(policy, replay, motion)["loss"])
                if self.record_batches:
                    self.batches.append((policy.clone(), replay.clone(), motion.clone()))
        return self._finish(rows, losses)

    def _finish(self, rows, losses, append=True):
        if append:
            self.replay.add_samples(rows)
        if getattr(self.trainer, "defer_refresh", False):
            self.trainer.refresh()  # the rollouts that follow score with the trained weights
        self._mark("train")
        return losses

    def _update_exchanged(self, rows, bs):
        """The multi-rank flow: this rank's ``bs / world`` rows of every group of every step -> ONE all-gather -> the steps."""
        from .distributed import UpdateExchange

        n = self.learning_epochs * self.mini_batches
        ex = self.exchange
        if ex is None or (ex.steps, ex.cols, ex.contrib.dtype, ex.contrib.device) != (n, rows.shape[1], rows.dtype, rows.device):
            ex = self.exchange = UpdateExchange(n, len(self.GROUPS), bs // _group_world(self.group), rows.shape[1], rows.device,
                                                dtype=rows.dtype, group=self.group)
        r = ex.rows_per_rank
        if rows.shape[0] < r * self.mini_batches:
            raise nat.AmpEngineError(f"the rollout has {rows.shape[0]} rows on this rank; {self.mini_batches} minibatches of {r} "
                                     f"(= {bs} / {ex.world} ranks) are needed")
        if len(self.motion_dataset) == 0:
            raise nat.AmpEngineError("the motion dataset is empty: fill it with collect_reference rows first")
        per = rows.shape[0] // self.mini_batches
        have_replay = len(self.replay) > 0
        epoch = self._epoch
        for k in range(n):
            mb = k % self.mini_batches
            if mb == 0:
                epoch, self._epoch = self._epoch, self._epoch + 1
            policy, replay, motion = ex.contrib[k].unbind(0)
            self._take_rows(rows, self.seed, epoch, mb * per, r, out=policy)
            if have_replay:
                self.replay.sample(r, out=replay, first_row=ex.first_row)
            else:
                replay.copy_(policy)
            self.motion_dataset.sample(r, out=motion, first_row=ex.first_row)
        self._mark("produce")
        ex.start()
        self.replay.add_samples(rows)  # the draws above are enqueued ahead of it; the append runs under the collective
        batches = ex.finish()
        self._mark("exchange")
        losses, self.batches = [], []
        for k in range(n):
            policy, replay, motion = batches[k].unbind(0)
            losses.append(self.trainer.step(policy, replay, motion)["loss"])
            if self.record_batches:
                self.batches.append((policy.clone(), replay.clone(), motion.clone()))
        return self._finish(rows, losses, append=False)

    def _update_prefetched(self, rows, per, bs):
        """The same flow with the batches of step k + 1 produced on a side stream under step k.  Two static buffer sets:
        the side stream refills a set only behind the training step that last read it (an event recorded on the main
        stream), the main stream trains on a set only behind its refill (an event recorded on the side stream).  The
        shuffles and the ring draws consume their counters in the in-line order.  The batches of step k + 1 are drawn before step k
        trains: if a training step raises, the epoch and ring draw counters are one step ahead of the in-line flow's (they are not
        rolled back).  The replay append follows the last draw on the side stream (see ``append`` below)."""
        dev = rows.device
        main = torch.cuda.current_stream(dev)
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        key = ((bs, rows.shape[1]), rows.dtype, dev)
        if self._bufs is None or self._bufs[0] != key:
            self._bufs = (key, [[torch.empty(key[0], dtype=rows.dtype, device=dev) for _ in range(3)] for _ in range(2)])
        side, bufs = self._side, self._bufs[1]
        n = self.learning_epochs * self.mini_batches
        ready, done = [None, None], [None, None]
        have_replay = len(self.replay) > 0
        state = {"epoch": self._epoch}
        side.wait_stream(main)  # the rollout rows (and whatever used the buffer sets before) are complete

        def produce(k):
            st, mb = k & 1, k % self.mini_batches
            with torch.cuda.stream(side):
                if done[st] is not None:
                    side.wait_event(done[st])
                if mb == 0:
                    state["epoch"], self._epoch = self._epoch, self._epoch + 1
                policy, replay, motion = bufs[st]
                self._take_rows(rows, self.seed, state["epoch"], mb * per, bs, out=policy)
                if have_replay:
                    self.replay.sample(bs, out=replay)
                else:
                    replay.copy_(policy)
                self.motion_dataset.sample(bs, out=motion)
                ready[st] = torch.cuda.Event()
                ready[st].record(side)

        def append():
            # the rollout's rows into the replay ring, on the side stream BEHIND the update's last ring draw (the same stream: ordered),
            # i.e. under the last training steps instead of after them (0.66 GB in + 0.66 GB out at the reference's sizes).  Like the draw
            # counters it is not rolled back if a later training step raises.
            with torch.cuda.stream(side):
                self.replay.add_samples(rows)

        losses = []
        produce(0)
        if n == 1:
            append()
        for k in range(n):
            if k + 1 < n:
                produce(k + 1)
                if k + 2 == n:
                    append()
            st = k & 1
            main.wait_event(ready[st])
            policy, replay, motion = bufs[st]
            losses.append(self.trainer.step(policy, replay, motion)["loss"])
            if self.record_batches:
                self.batches.append((policy.clone(), replay.clone(), motion.clone()))
            done[st] = torch.cuda.Event()
            done[st].record(main)
        main.wait_stream(side)   # the ring holds the rollout's rows for whatever follows on the caller's stream
        return self._finish(rows, losses, append=False)


def _group_world(group) -> int:
    import torch.distributed as dist

    return dist.get_world_size(group)
