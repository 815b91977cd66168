"""MotionLoader: drop-in surface of the reference's ``motions/motion_loader.py`` on the HIP engine.

Same constructor, attributes and methods (reference: motions/motion_loader.py:87-430); the numerics run in
``libamp_engine.so`` (csrc/motion.hip):

    sample()             -> amp_motion_sample        (fp64 frame/blend index + 5x LERP + SLERP, one launch
                                                      instead of ~80 ATen launches and 13 H2D copies)
    _compute_frame_blend -> amp_motion_frame_blend
    collect_reference()  -> amp_collect_reference    (engine-only fast path used by the envs)

Host-only parts (path mini-language, name lookups, the numpy RNG of ``sample_times``) are plain Python and
work on any device; anything that computes needs a HIP device and raises otherwise.
"""

from __future__ import annotations

import ctypes as C
import glob
import os
from typing import Optional, Sequence

import numpy as np
import torch
import yaml

from .. import _native as nat

_TABLES = ("dof_positions", "dof_velocities", "body_positions", "body_rotations",
           "body_linear_velocities", "body_angular_velocities")


def _resolve_motion_files(motion_file: str) -> list[str]:
    """Resolve the reference's path mini-language (motions/motion_loader.py:14-84), tried in this order:
    YAML config (``motion_files`` list, else ``glob_pattern``) | comma list | glob | directory | file."""
    if motion_file.endswith((".yaml", ".yml")):
        root = os.path.dirname(motion_file)
        with open(motion_file, "r") as fh:
            cfg = yaml.safe_load(fh)
        anchored = lambda p: p if os.path.isabs(p) else os.path.join(root, p)  # noqa: E731
        found: list[str] = []
        if cfg and "motion_files" in cfg:
            for entry in cfg["motion_files"]:
                path = anchored(entry)
                if os.path.exists(path):
                    found.append(path)
                else:
                    print(f"Warning: File not found: {path}")
        if not found and "glob_pattern" in cfg:
            found = sorted(glob.glob(anchored(cfg["glob_pattern"])))
        if not found:
            raise ValueError(f"No valid motion files found in config: {motion_file}")
        return found
    if "," in motion_file:
        listed = [p.strip() for p in motion_file.split(",")]
        listed = [p for p in listed if os.path.exists(p)]
        if listed:
            return listed
    if any(ch in motion_file for ch in "*?"):
        matched = sorted(glob.glob(motion_file))
        if matched:
            return matched
    if os.path.isdir(motion_file):
        inside = sorted(glob.glob(os.path.join(motion_file, "*.npz")))
        if inside:
            return inside
    if os.path.exists(motion_file):
        return [motion_file]
    raise ValueError(f"No files found for pattern: {motion_file}")


def _as_device(x, dtype, device, n: Optional[int] = None, name: str = "array") -> torch.Tensor:
    """numpy / list / tensor -> contiguous device tensor of ``dtype`` (one H2D copy for host inputs)."""
    if isinstance(x, torch.Tensor):
        t = x.to(device=device, dtype=dtype).contiguous()
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)).astype(_NP[dtype], copy=False)).to(device)
    if t.dim() != 1 or (n is not None and t.numel() != n):
        raise ValueError(f"{name} must be 1-D" + (f" with {n} entries, got {tuple(t.shape)}" if n is not None else ""))
    return t


_NP = {torch.float64: np.float64, torch.int64: np.int64}


class MotionLoader:
    """Load 1..n ``.npz`` clips and sample them by time (LERP / SLERP) on the MI355X."""

    def __init__(self, motion_file: str, device) -> None:
        files = _resolve_motion_files(motion_file)
        print(f"Loading {len(files)} motion file(s) from: {motion_file}")
        self.device = device
        self._tdev = torch.device(device)
        self.num_trajectories = len(files)

        chunks = {k: [] for k in _TABLES}
        frames = []
        for i, path in enumerate(files):
            with np.load(path) as clip:
                if i == 0:  # names and dt come from the first clip only (motion_loader.py:119-122)
                    self._dof_names = clip["dof_names"].tolist()
                    self._body_names = clip["body_names"].tolist()
                    self.dt = 1.0 / clip["fps"]
                for k in _TABLES:
                    chunks[k].append(clip[k])
                frames.append(int(clip["dof_positions"].shape[0]))
        frames_np = np.asarray(frames, dtype=np.int64)
        ends = np.cumsum(frames_np)
        self.traj_starts = ends - frames_np
        self.traj_ends = ends - 1
        self.durations = np.array([self.dt * (n - 1) for n in frames])
        for k in _TABLES:  # float64 clips (G1_walk) are cast to fp32 exactly like the reference does
            setattr(self, k, torch.tensor(np.concatenate(chunks[k]), dtype=torch.float32, device=self._tdev))
        self.num_frames = int(ends[-1])
        self.duration = float(np.sum(self.durations))
        self._clip_frames = frames_np
        self._handle = None
        self._layout = None
        if self._tdev.type == "cuda":
            self._create_handle()
        print(f"Motion loaded: {self.num_trajectories} files, total duration: {self.duration} sec, "
              f"total frames: {self.num_frames}")

    # ---- native handle ----------------------------------------------------------------------------

    def _create_handle(self) -> None:
        lib = nat.load()
        nat.require_gpu(self._tdev)
        if int(self._clip_frames.min()) < 2:
            raise ValueError("every motion clip needs at least 2 frames")
        d = nat.AmpMotionDesc()
        d.n_clips, d.n_dof, d.n_bodies = self.num_trajectories, self.num_dofs, self.num_bodies
        d.n_frames, d.dt = self.num_frames, float(self.dt)
        self._clip_frames_c = (C.c_int64 * self.num_trajectories)(*self._clip_frames.tolist())
        d.clip_frames = self._clip_frames_c
        for k in _TABLES:
            setattr(d, k, getattr(self, k).data_ptr())
        h = C.c_void_p()
        with torch.cuda.device(self._tdev):
            nat.check(lib.amp_motion_create(C.byref(d), C.byref(h)), "amp_motion_create")
        self._handle = h
        self._lib = lib

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h is not None:
            try:
                self._lib.amp_motion_destroy(h)
            except Exception:
                pass

    def _need_handle(self):
        if self._handle is None:
            nat.require_gpu(self._tdev)  # raises: no CPU path
        return self._handle

    # ---- reference surface -------------------------------------------------------------------------

    @property
    def dof_names(self) -> list[str]:
        return self._dof_names

    @property
    def body_names(self) -> list[str]:
        return self._body_names

    @property
    def num_dofs(self) -> int:
        return len(self._dof_names)

    @property
    def num_bodies(self) -> int:
        return len(self._body_names)

    def get_dof_index(self, dof_names: Sequence[str]) -> list[int]:
        out = []
        for name in dof_names:
            assert name in self._dof_names, f"The specified DOF name ({name}) doesn't exist: {self._dof_names}"
            out.append(self._dof_names.index(name))
        return out

    def get_body_index(self, body_names: Sequence[str]) -> list[int]:
        out = []
        for name in body_names:
            assert name in self._body_names, f"The specified body name ({name}) doesn't exist: {self._body_names}"
            out.append(self._body_names.index(name))
        return out

    def sample_times(self, num_samples: int, start: bool = False) -> tuple[np.ndarray, np.ndarray]:
        """(motion_ids, times) from the global legacy numpy RNG, as the reference (motion_loader.py:309-329)."""
        motion_ids = np.random.randint(0, self.num_trajectories, size=num_samples)
        if start:
            return motion_ids, np.zeros(num_samples)
        return motion_ids, np.random.uniform(low=0.0, high=1.0, size=num_samples) * self.durations[motion_ids]

    def _check_ids(self, motion_ids) -> None:
        if isinstance(motion_ids, torch.Tensor):
            return  # device ids are clamped in-kernel; validating them would force a sync
        ids = np.asarray(motion_ids)
        if ids.size and (ids.min() < 0 or ids.max() >= self.num_trajectories):
            raise IndexError(f"motion_ids out of range [0, {self.num_trajectories})")

    def _compute_frame_blend(self, times, motion_ids):
        """(index_0, index_1, blend) of motion_loader.py:281-307, computed by ``amp_motion_frame_blend``.  Return types
        follow the input: numpy ``times`` (the reference's calling convention) -> numpy int64 / int64 / float64 arrays like
        the reference returns; device tensors in -> device tensors out (no host round trip)."""
        as_numpy = not isinstance(times, torch.Tensor)
        h = self._need_handle()
        t = _as_device(times, torch.float64, self._tdev, name="times")
        n = t.numel()
        self._check_ids(motion_ids)
        ids = _as_device(motion_ids, torch.int64, self._tdev, n, "motion_ids")
        i0 = torch.empty(n, dtype=torch.int64, device=self._tdev)
        i1 = torch.empty_like(i0)
        blend = torch.empty(n, dtype=torch.float64, device=self._tdev)
        with torch.cuda.device(self._tdev):
            nat.check(self._lib.amp_motion_frame_blend(h, nat.dptr(t), nat.dptr(ids), n, nat.dptr(i0), nat.dptr(i1),
                                                       nat.dptr(blend), nat.stream_ptr()), "amp_motion_frame_blend")
        if as_numpy:
            return i0.cpu().numpy(), i1.cpu().numpy(), blend.cpu().numpy()
        return i0, i1, blend

    def sample(self, num_samples: int, times=None, duration: float | None = None, motion_ids=None):
        """DOF pos, DOF vel, body pos, body rot (wxyz), body lin vel, body ang vel -- motion_loader.py:331-390.

        ``times`` / ``motion_ids`` may be numpy arrays (reference convention; one H2D copy each) or device
        tensors (float64 / int64; no copy, no sync).  ``duration`` is accepted and ignored, as in the reference.
        """
        h = self._need_handle()
        if times is None:
            new_ids, times = self.sample_times(num_samples)
            if motion_ids is None:
                motion_ids = new_ids
        t = _as_device(times, torch.float64, self._tdev, name="times")
        n = t.numel()
        ids = None
        if motion_ids is not None:  # None -> clip 0 for every sample (motion_loader.py:365-366)
            self._check_ids(motion_ids)
            ids = _as_device(motion_ids, torch.int64, self._tdev, n, "motion_ids")
        dev, B, nd = self._tdev, self.num_bodies, self.num_dofs
        f32 = dict(dtype=torch.float32, device=dev)
        out = (torch.empty((n, nd), **f32), torch.empty((n, nd), **f32), torch.empty((n, B, 3), **f32),
               torch.empty((n, B, 4), **f32), torch.empty((n, B, 3), **f32), torch.empty((n, B, 3), **f32))
        with torch.cuda.device(dev):
            nat.check(self._lib.amp_motion_sample(h, nat.dptr(t), nat.dptr(ids), n, *[nat.dptr(o) for o in out],
                                                  nat.stream_ptr()), "amp_motion_sample")
        return out

    # ---- engine fast paths used by the envs -----------------------------------------------------------

    def set_obs_layout(self, dof_indexes: Sequence[int], ref_body_index: int, key_body_indexes: Sequence[int]) -> int:
        """Fix the robot-order DoF permutation, reference body and key bodies of the AMP observation
        (g1_amp_env.py:47-60); returns the per-frame observation size D."""
        h = self._need_handle()
        if len(dof_indexes) != self.num_dofs:
            raise ValueError(f"dof_indexes must have {self.num_dofs} entries")
        perm = (C.c_int32 * self.num_dofs)(*[int(i) for i in dof_indexes])
        keys = (C.c_int32 * len(key_body_indexes))(*[int(i) for i in key_body_indexes])
        with torch.cuda.device(self._tdev):
            nat.check(self._lib.amp_motion_set_obs_layout(h, perm, int(ref_body_index), keys, len(key_body_indexes),
                                                          nat.stream_ptr()), "amp_motion_set_obs_layout")
        self._layout = (list(dof_indexes), int(ref_body_index), list(key_body_indexes))
        return 2 * self.num_dofs + 13 + 3 * len(key_body_indexes)

    @property
    def obs_size(self) -> int:
        if self._layout is None:
            raise nat.AmpEngineError("call set_obs_layout first")
        return 2 * self.num_dofs + 13 + 3 * len(self._layout[2])

    def collect_reference(self, times, motion_ids, num_amp_observations: int, out: torch.Tensor | None = None,
                          dst_rows: torch.Tensor | None = None) -> torch.Tensor:
        """Expert AMP observations [n, K*D] for (times, ids): K frames t - dt*k, newest first
        (g1_amp_env.py:445-486 fused with compute_obs :535-561).  With ``out`` ([N,K,D] buffer) and ``dst_rows``
        (int64 env ids) the rows are scattered in place: ``out[dst_rows] = expert rows`` (g1_amp_env.py:417)."""
        h = self._need_handle()
        K, D = int(num_amp_observations), self.obs_size
        t = _as_device(times, torch.float64, self._tdev, name="times")
        n = t.numel()
        ids = None
        if motion_ids is not None:
            self._check_ids(motion_ids)
            ids = _as_device(motion_ids, torch.int64, self._tdev, n, "motion_ids")
        if out is None:
            if dst_rows is not None:
                raise ValueError("dst_rows needs an explicit out buffer")
            out = torch.empty((n, K * D), dtype=torch.float32, device=self._tdev)
        elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() % (K * D) != 0:
            raise ValueError(f"out must be a contiguous float32 buffer of rows of {K}*{D} floats")
        rows = None
        if dst_rows is not None:
            rows = _as_device(dst_rows, torch.int64, self._tdev, n, "dst_rows")
        elif out.numel() < n * K * D:
            raise ValueError("out is too small")
        with torch.cuda.device(self._tdev):
            nat.check(self._lib.amp_collect_reference(h, nat.dptr(t), nat.dptr(ids), n, K, nat.dptr(out), nat.dptr(rows),
                                                      nat.stream_ptr()), "amp_collect_reference")
        return out

    def sample_times_device(self, num_samples: int, start: bool = False, *, seed: int = 0, step: int = 0,
                            index: torch.Tensor | None = None, count: torch.Tensor | None = None):
        """Device-side :meth:`sample_times`: (motion_ids int64 [n], times float64 [n]) drawn with the counter-based
        Philox generator keyed by ``(seed, step, index[i] or i)`` -- no host RNG, no H2D copy.  ``count`` (device int64
        [1]) caps the draws without a read-back.  Distribution as the reference; the random stream is the engine's own."""
        h = self._need_handle()
        n = int(num_samples)
        ids = torch.zeros(n, dtype=torch.int64, device=self._tdev)
        times = torch.zeros(n, dtype=torch.float64, device=self._tdev)
        with torch.cuda.device(self._tdev):
            nat.check(self._lib.amp_motion_sample_times(h, int(seed) & (2**64 - 1), int(step) & (2**64 - 1), int(bool(start)),
                                                        nat.dptr(index, torch.int64, "index"), nat.dptr(count, torch.int64, "count"),
                                                        n, nat.dptr(ids), nat.dptr(times), nat.stream_ptr()),
                      "amp_motion_sample_times")
        return ids, times

    def reset_apply(self, env_ids: torch.Tensor, count: torch.Tensor, num_amp_observations: int, *, seed: int, step: int,
                    start: bool, env_origins: torch.Tensor | None, z_lift: float, amp_observation_buffer: torch.Tensor,
                    out: dict | None = None, env_motion_ids: torch.Tensor | None = None,
                    env_motion_start_times: torch.Tensor | None = None, env_offset: int = 0) -> dict:
        """The whole reference-state reset (g1_amp_env.py:371-419) on the device-side output of the reset compaction:
        for i < count, env = env_ids[i]: draw (clip, t), write root_state[i] / dof_pos[i] / dof_vel[i] and the K expert
        frames into ``amp_observation_buffer[env]``.  Returns the (reusable) compact output tensors.
        ``env_motion_ids`` / ``env_motion_start_times`` ([num_envs] int64 / float32) receive the draw per env (the env's
        ``motion_ids`` / ``motion_start_times`` attributes); draws are keyed by ``env_offset + env`` (shard-invariant)."""
        h = self._need_handle()
        if self._layout is None:
            raise nat.AmpEngineError("call set_obs_layout first")
        n = env_ids.numel()
        if out is None:
            f32 = dict(dtype=torch.float32, device=self._tdev)
            out = dict(root_state=torch.zeros((n, 13), **f32), dof_pos=torch.zeros((n, self.num_dofs), **f32),
                       dof_vel=torch.zeros((n, self.num_dofs), **f32), motion_ids=torch.zeros(n, dtype=torch.int64, device=self._tdev),
                       motion_times=torch.zeros(n, dtype=torch.float64, device=self._tdev))
        a = nat.AmpResetArgs()
        a.env_ids, a.count, a.max_n = nat.dptr(env_ids, torch.int64, "env_ids").value, nat.dptr(count, torch.int64, "count").value, n
        a.seed, a.step, a.start, a.K = int(seed) & (2**64 - 1), int(step) & (2**64 - 1), int(bool(start)), int(num_amp_observations)
        a.env_origins = nat.dptr(env_origins, torch.float32, "env_origins").value
        a.z_lift = float(z_lift)
        a.root_state, a.dof_pos, a.dof_vel = (out[k].data_ptr() for k in ("root_state", "dof_pos", "dof_vel"))
        a.amp_obs_buffer = nat.dptr(amp_observation_buffer, torch.float32, "amp_observation_buffer").value
        a.motion_ids, a.motion_times = out["motion_ids"].data_ptr(), out["motion_times"].data_ptr()
        a.env_motion_ids = nat.dptr(env_motion_ids, torch.int64, "env_motion_ids").value
        a.env_motion_start_times = nat.dptr(env_motion_start_times, torch.float32, "env_motion_start_times").value
        a.env_offset = int(env_offset)
        with torch.cuda.device(self._tdev):
            nat.check(self._lib.amp_reset_apply(h, C.byref(a), nat.stream_ptr()), "amp_reset_apply")
        return out

    def reset_compact_apply(self, reset_mask: torch.Tensor, tile_counts: torch.Tensor, tile_envs: int, env_ids: torch.Tensor,
                            count: torch.Tensor, num_amp_observations: int, *, seed: int, step: int, start: bool,
                            env_origins: torch.Tensor | None, z_lift: float, amp_observation_buffer: torch.Tensor | None,
                            out: dict | None = None, env_motion_ids: torch.Tensor | None = None,
                            env_motion_start_times: torch.Tensor | None = None, env_offset: int = 0,
                            episode_length: torch.Tensor | None = None, last_actions: torch.Tensor | None = None,
                            just_reset: torch.Tensor | None = None, command: "nat.AmpCommandArgs | None" = None,
                            reward_terms: torch.Tensor | None = None, reward_means: torch.Tensor | None = None) -> dict:
        """Reset-id compaction (``reset_mask`` + the per-tile counts of the DONES launch -> ``env_ids`` / ``count``, both
        written here) and :meth:`reset_apply` on them as ONE launch (``amp_reset_compact_apply``), plus the optional
        per-env clears (``episode_length[env] = 0``, ``last_actions[env] = 0``, ``just_reset[env] = True``) and the
        reset-side command resample (``command``: a filled ``AmpCommandArgs``).  Bit-identical to the separate calls."""
        h = self._need_handle()
        if self._layout is None:
            raise nat.AmpEngineError("call set_obs_layout first")
        n = int(reset_mask.numel())
        if env_ids.numel() < n:
            raise nat.AmpEngineError("env_ids must have room for every env")
        if out is None:
            f32 = dict(dtype=torch.float32, device=self._tdev)
            out = dict(root_state=torch.zeros((n, 13), **f32), dof_pos=torch.zeros((n, self.num_dofs), **f32),
                       dof_vel=torch.zeros((n, self.num_dofs), **f32), motion_ids=torch.zeros(n, dtype=torch.int64, device=self._tdev),
                       motion_times=torch.zeros(n, dtype=torch.float64, device=self._tdev))
        c = nat.AmpCompactArgs()
        c.mask, c.tile_counts = nat.dptr(reset_mask, None, "reset_mask").value, nat.dptr(tile_counts, torch.int32, "tile_counts").value
        c.tile_envs, c.num_envs = int(tile_envs), n
        c.ids, c.count = nat.dptr(env_ids, torch.int64, "env_ids").value, nat.dptr(count, torch.int64, "count").value
        a = nat.AmpResetArgs()
        a.env_ids, a.count, a.max_n = c.ids, c.count, int(env_ids.numel())
        a.seed, a.step, a.start, a.K = int(seed) & (2**64 - 1), int(step) & (2**64 - 1), int(bool(start)), int(num_amp_observations)
        a.env_origins = nat.dptr(env_origins, torch.float32, "env_origins").value
        a.z_lift = float(z_lift)
        a.root_state, a.dof_pos, a.dof_vel = (out[k].data_ptr() for k in ("root_state", "dof_pos", "dof_vel"))
        a.amp_obs_buffer = nat.dptr(amp_observation_buffer, torch.float32, "amp_observation_buffer").value
        a.motion_ids, a.motion_times = out["motion_ids"].data_ptr(), out["motion_times"].data_ptr()
        a.env_motion_ids = nat.dptr(env_motion_ids, torch.int64, "env_motion_ids").value
        a.env_motion_start_times = nat.dptr(env_motion_start_times, torch.float32, "env_motion_start_times").value
        a.env_offset = int(env_offset)
        a.episode_length = nat.dptr(episode_length, torch.int64, "episode_length").value
        if last_actions is not None:
            a.last_actions, a.n_actions = nat.dptr(last_actions, torch.float32, "last_actions").value, int(last_actions.shape[1])
        a.just_reset = nat.dptr(just_reset, None, "just_reset").value
        lg = None
        if reward_terms is not None:  # the step's reward-log means on the same launch -> reward_means [n_terms]
            lg = nat.AmpRewardLogArgs()
            lg.reward_terms, lg.n_terms = nat.dptr(reward_terms, torch.float32, "reward_terms").value, int(reward_terms.shape[0])
            lg.means = nat.dptr(reward_means, torch.float32, "reward_means").value
        with torch.cuda.device(self._tdev):
            nat.check(self._lib.amp_reset_compact_apply(h, C.byref(c), C.byref(a), C.byref(command) if command is not None else None,
                                                        C.byref(lg) if lg is not None else None, nat.stream_ptr()),
                      "amp_reset_compact_apply")
        return out

    def reset_reference_state(self, times, motion_ids, env_ids=None, env_origins: torch.Tensor | None = None,
                              z_lift: float = 0.0):
        """(root_state [n,13], dof_pos [n,Dof], dof_vel [n,Dof]) of the reference body for reset envs
        (g1_amp_env.py:385-411): position + env origin, z lifted; robot-order DoFs."""
        h = self._need_handle()
        if self._layout is None:
            raise nat.AmpEngineError("call set_obs_layout first")
        t = _as_device(times, torch.float64, self._tdev, name="times")
        n = t.numel()
        ids = None
        if motion_ids is not None:
            self._check_ids(motion_ids)
            ids = _as_device(motion_ids, torch.int64, self._tdev, n, "motion_ids")
        eids = None if env_ids is None else _as_device(env_ids, torch.int64, self._tdev, n, "env_ids")
        f32 = dict(dtype=torch.float32, device=self._tdev)
        root = torch.empty((n, 13), **f32)
        dpos = torch.empty((n, self.num_dofs), **f32)
        dvel = torch.empty((n, self.num_dofs), **f32)
        with torch.cuda.device(self._tdev):
            nat.check(self._lib.amp_reset_reference_state(h, nat.dptr(t), nat.dptr(ids), nat.dptr(eids), n,
                                                          nat.dptr(env_origins, torch.float32, "env_origins"),
                                                          float(z_lift), nat.dptr(root), nat.dptr(dpos), nat.dptr(dvel),
                                                          nat.stream_ptr()), "amp_reset_reference_state")
        return root, dpos, dvel


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--file", type=str, required=True, help="Motion file")
    ns, _ = ap.parse_known_args()
    m = MotionLoader(ns.file, "cuda:0" if torch.cuda.is_available() else "cpu")
    for label, value in (("number of frames", m.num_frames), ("number of DOFs", m.num_dofs), ("dt", m.dt),
                         ("fps", 1.0 / m.dt), ("number of bodies", m.num_bodies)):
        print(f"- {label}:", value)
