"""ctypes binding of ``libamp_engine.so`` (C ABI declared in ``include/amp_engine.h``).

There is NO fallback: if the HIP library is missing or a call fails, this module raises.  torch is used
only as plumbing (device memory, current stream).
"""

from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: its libamdhip64.so.7 is the HIP runtime we bind to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMP_ENGINE_LIB: load another build of the SAME library instead of the in-tree one (A/B variants from tools/build_variant.sh:
# diagnostic stamps, ablations) -- the in-tree product file is never overwritten by an experiment.  ABI and symbols are checked as usual.
LIB_PATH = os.environ.get("AMP_ENGINE_LIB") or os.path.join(_HERE, "csrc", "libamp_engine.so")
ABI_VERSION = 11

AMP_DISC_F16X3, AMP_DISC_FP32 = 0, 1
AMP_DISC_INPUT_F32_ROWS, AMP_DISC_INPUT_F16_BLOCKS = 0, 1
AMP_PHASE_DONES, AMP_PHASE_REWARD, AMP_PHASE_OBS = 1, 2, 4
AMP_PHASE_ALL = 7
AMP_COMMAND_TICK, AMP_COMMAND_RESET = 0, 1
AMP_RESET_REFERENCE, AMP_RESET_DEFAULT = 0, 1
TILE_ENVS = 64


class AmpEngineError(RuntimeError):
    """A libamp_engine.so entry point returned an error code."""


class AmpMotionDesc(C.Structure):
    _fields_ = [
        ("n_clips", C.c_int32), ("n_dof", C.c_int32), ("n_bodies", C.c_int32), ("reserved", C.c_int32),
        ("n_frames", C.c_int64), ("dt", C.c_double), ("clip_frames", C.POINTER(C.c_int64)),
        ("dof_positions", C.c_void_p), ("dof_velocities", C.c_void_p), ("body_positions", C.c_void_p),
        ("body_rotations", C.c_void_p), ("body_linear_velocities", C.c_void_p), ("body_angular_velocities", C.c_void_p),
    ]


class AmpResetArgs(C.Structure):
    _fields_ = [
        ("env_ids", C.c_void_p), ("count", C.c_void_p), ("max_n", C.c_int64), ("seed", C.c_uint64), ("step", C.c_uint64),
        ("start", C.c_int32), ("K", C.c_int32), ("env_origins", C.c_void_p), ("z_lift", C.c_float), ("mode", C.c_int32),
        ("root_state", C.c_void_p), ("dof_pos", C.c_void_p), ("dof_vel", C.c_void_p), ("amp_obs_buffer", C.c_void_p),
        ("motion_ids", C.c_void_p), ("motion_times", C.c_void_p),
        ("env_motion_ids", C.c_void_p), ("env_motion_start_times", C.c_void_p), ("env_offset", C.c_int64),
        ("episode_length", C.c_void_p), ("last_actions", C.c_void_p), ("just_reset", C.c_void_p), ("n_actions", C.c_int32),
        ("reserved2", C.c_int32), ("step_dev", C.c_void_p),
        ("default_root_state", C.c_void_p), ("default_joint_pos", C.c_void_p), ("default_joint_vel", C.c_void_p),
        ("step_dev_out", C.c_void_p),
    ]


class AmpCompactArgs(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("tile_counts", C.c_void_p), ("tile_envs", C.c_int32), ("reserved", C.c_int32),
                ("num_envs", C.c_int64), ("ids", C.c_void_p), ("count", C.c_void_p)]


class AmpPrePhysicsArgs(C.Structure):
    _fields_ = [("actions_in", C.c_void_p), ("actions", C.c_void_p), ("last_actions", C.c_void_p), ("target", C.c_void_p),
                ("offset", C.c_void_p), ("scale", C.c_void_p), ("num_envs", C.c_int64), ("n_actions", C.c_int32), ("reserved", C.c_int32),
                ("episode_length", C.c_void_p), ("step_in", C.c_void_p), ("step_out", C.c_void_p)]


class AmpRewardLogArgs(C.Structure):
    _fields_ = [("reward_terms", C.c_void_p), ("n_terms", C.c_int32), ("reserved", C.c_int32), ("means", C.c_void_p)]


class AmpScatterRows(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_stride", C.c_int64), ("fill", C.c_float), ("width", C.c_int32), ("repeat", C.c_int32),
                ("reserved", C.c_int32), ("add", C.c_void_p), ("dst", C.c_void_p), ("dst_stride", C.c_int64)]


class AmpHotStepArgs(C.Structure):
    _fields_ = [("cfg", C.c_void_p), ("state", C.c_void_p), ("bufs", C.c_void_p), ("num_envs", C.c_int64), ("motion", C.c_void_p),
                ("times", C.c_void_p), ("motion_ids", C.c_void_p), ("n_samples", C.c_int64), ("K", C.c_int32), ("reserved", C.c_int32),
                ("expert_out", C.c_void_p), ("disc", C.c_void_p), ("reward_scale", C.c_float), ("task_weight", C.c_float),
                ("style_weight", C.c_float), ("reserved2", C.c_int32), ("logits", C.c_void_p), ("style", C.c_void_p),
                ("combined", C.c_void_p), ("workspace", C.c_void_p), ("compact", C.c_void_p)]


class AmpCommandArgs(C.Structure):
    _fields_ = [
        ("command", C.c_void_p), ("time_left", C.c_void_p), ("step_dt", C.c_float), ("vel_lo", C.c_float), ("vel_span", C.c_float),
        ("t_lo", C.c_float), ("t_span", C.c_float), ("reserved", C.c_int32), ("seed", C.c_uint64), ("step", C.c_uint64),
        ("env_offset", C.c_int64), ("reset_mask", C.c_void_p), ("env_ids", C.c_void_p), ("count", C.c_void_p), ("n_ids", C.c_int64),
        ("step_dev", C.c_void_p),
    ]


class AmpDiscTrainCfg(C.Structure):
    _fields_ = [
        ("max_rows_per_group", C.c_int64), ("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
        ("adam_epsilon", C.c_float), ("loss_scale", C.c_float), ("logit_reg_scale", C.c_float), ("grad_penalty_scale", C.c_float),
        ("weight_decay_scale", C.c_float), ("scaler_epsilon", C.c_float), ("scaler_clip", C.c_float), ("use_scaler", C.c_int32),
        ("update_scaler", C.c_int32), ("apply_update", C.c_int32), ("gemm_f16x3", C.c_int32), ("defer_refresh", C.c_int32), ("reserved", C.c_int32),
    ]


class AmpEnvCfg(C.Structure):
    _fields_ = [
        ("n_dof", C.c_int32), ("n_key", C.c_int32), ("num_amp_observations", C.c_int32),
        ("num_actor_observations", C.c_int32), ("use_last_actions", C.c_int32), ("use_command", C.c_int32),
        ("history_include_last_actions", C.c_int32), ("history_include_command", C.c_int32),
        ("early_termination", C.c_int32), ("reward_mode", C.c_int32), ("max_episode_length", C.c_int64),
        ("termination_height", C.c_float), ("rew_termination", C.c_float), ("rew_action_l2", C.c_float),
        ("rew_joint_pos_limits", C.c_float), ("rew_joint_acc_l2", C.c_float), ("rew_joint_vel_l2", C.c_float),
        ("rew_track_vel", C.c_double), ("track_sigma", C.c_double), ("track_floor", C.c_double),
    ]


class AmpSimState(C.Structure):
    _fields_ = [
        ("joint_pos", C.c_void_p), ("joint_pos_stride", C.c_int64),
        ("joint_vel", C.c_void_p), ("joint_vel_stride", C.c_int64),
        ("joint_acc", C.c_void_p), ("joint_acc_stride", C.c_int64),
        ("actions", C.c_void_p), ("actions_stride", C.c_int64),
        ("root_pos", C.c_void_p), ("root_pos_stride", C.c_int64),
        ("root_quat", C.c_void_p), ("root_quat_stride", C.c_int64),
        ("root_lin_vel", C.c_void_p), ("root_lin_vel_stride", C.c_int64),
        ("root_ang_vel", C.c_void_p), ("root_ang_vel_stride", C.c_int64),
        ("body_pos", C.c_void_p), ("body_pos_stride", C.c_int64),
        ("key_body", C.c_int32 * 8),
        ("soft_limits", C.c_void_p), ("soft_limits_stride", C.c_int64),
        ("episode_length", C.c_void_p), ("command", C.c_void_p), ("last_actions", C.c_void_p),
    ]


class AmpEnvBuffers(C.Structure):
    _fields_ = [
        ("amp_obs_buffer", C.c_void_p), ("policy_obs", C.c_void_p), ("actor_history", C.c_void_p),
        ("just_reset", C.c_void_p), ("reward", C.c_void_p), ("reward_terms", C.c_void_p), ("died", C.c_void_p),
        ("time_out", C.c_void_p), ("reset_mask", C.c_void_p), ("reset_tile_counts", C.c_void_p),
        ("disc_input", C.c_void_p), ("disc_input_stride", C.c_int64), ("scaler_mean", C.c_void_p), ("scaler_den", C.c_void_p),
        ("scaler_clip", C.c_float), ("disc_input_format", C.c_int32), ("disc_plane_scale", C.c_float),
        ("amp_obs_read_next", C.c_int32),
    ]


class AmpKinModel(C.Structure):
    _fields_ = [("n_joints", C.c_int32), ("n_dof", C.c_int32), ("n_bodies", C.c_int32), ("reserved", C.c_int32),
                ("parent", C.c_void_p), ("qidx", C.c_void_p), ("origin_rot", C.c_void_p), ("origin_xyz", C.c_void_p),
                ("axis", C.c_void_p), ("body_joint", C.c_void_p)]


class AmpConvertOutputs(C.Structure):
    _fields_ = [("dof_positions", C.c_void_p), ("dof_velocities", C.c_void_p), ("body_positions", C.c_void_p),
                ("body_rotations", C.c_void_p), ("body_linear_velocities", C.c_void_p), ("body_angular_velocities", C.c_void_p)]


class AmpDiscInputLayout(C.Structure):
    _fields_ = [("format", C.c_int32), ("padded_dim", C.c_int32), ("mean_dev", C.c_void_p), ("den_dev", C.c_void_p),
                ("clip", C.c_float), ("plane_scale", C.c_float)]


class AmpDiscPlanInfo(C.Structure):
    _fields_ = [("precision", C.c_int32), ("plan", C.c_int32), ("fused_rows", C.c_int64), ("chunk_rows", C.c_int64),
                ("fused_min_rows", C.c_int64), ("env_overrides", C.c_int32), ("cu_count", C.c_int32), ("raw_input", C.c_int32),
                ("reserved", C.c_int32)]


class AmpDiscDesc(C.Structure):
    _fields_ = [
        ("in_dim", C.c_int32), ("h1", C.c_int32), ("h2", C.c_int32), ("reserved", C.c_int32),
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p),
    ]


_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float

# name -> (restype, argtypes); every symbol include/amp_engine.h declares
SIGNATURES = {
    "amp_abi_version": (C.c_int, []),
    "amp_last_error": (C.c_char_p, []),
    "amp_device_name": (C.c_int, [C.c_char_p, _i64]),
    "amp_trace_begin": (C.c_int, [_i64, C.c_char_p]),
    "amp_trace_sample": (C.c_int, [_i64]),
    "amp_trace_end": (C.c_int, []),
    "amp_trace_count": (_i64, []),
    "amp_trace_get": (C.c_int, [_i64, C.c_char_p, _i64, C.POINTER(C.c_float)]),
    "amp_calibrate_mfma_f16": (C.c_int, [_i32, _i32, _vp, _i64, C.POINTER(C.c_double), _vp]),
    "amp_motion_create": (C.c_int, [C.POINTER(AmpMotionDesc), C.POINTER(_vp)]),
    "amp_motion_destroy": (C.c_int, [_vp]),
    "amp_motion_set_obs_layout": (C.c_int, [_vp, C.POINTER(_i32), _i32, C.POINTER(_i32), _i32, _vp]),
    "amp_motion_frame_blend": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "amp_motion_sample": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amp_collect_reference": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "amp_reset_reference_state": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _vp, _vp, _vp, _vp]),
    "amp_motion_sample_times": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    "amp_reset_apply": (C.c_int, [_vp, C.POINTER(AmpResetArgs), _vp]),
    "amp_reset_compact_apply": (C.c_int, [_vp, C.POINTER(AmpCompactArgs), C.POINTER(AmpResetArgs), C.POINTER(AmpCommandArgs),
                                          C.POINTER(AmpRewardLogArgs), _vp]),
    "amp_policy_obs_size": (_i64, [C.POINTER(AmpEnvCfg)]),
    "amp_actor_history_frame_size": (_i64, [C.POINTER(AmpEnvCfg)]),
    "amp_env_step": (C.c_int, [C.POINTER(AmpEnvCfg), C.POINTER(AmpSimState), C.POINTER(AmpEnvBuffers), _i64, C.c_uint32, _vp]),
    "amp_env_step_with_reference": (C.c_int, [C.POINTER(AmpEnvCfg), C.POINTER(AmpSimState), C.POINTER(AmpEnvBuffers), _i64, C.c_uint32,
                                              _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "amp_command_step": (C.c_int, [C.POINTER(AmpCommandArgs), _i64, _i32, _vp]),
    "amp_pre_physics_step": (C.c_int, [C.POINTER(AmpPrePhysicsArgs), C.POINTER(AmpCommandArgs), _vp]),
    "amp_reward_log_means": (C.c_int, [_vp, _i32, _i64, _vp, _vp]),
    "amp_reset_compact_workspace_bytes": (_i64, [_i64]),
    "amp_reset_compact": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "amp_reset_compact_tiles": (C.c_int, [_vp, _vp, _i32, _i64, _vp, _vp, _vp]),
    "amp_scatter_rows": (C.c_int, [C.POINTER(AmpScatterRows), _i32, _vp, _vp, _i64, _vp]),
    "amp_env_step_tile_envs": (_i32, [C.POINTER(AmpEnvCfg), _i64]),
    "amp_disc_create": (C.c_int, [C.POINTER(AmpDiscDesc), _vp, C.POINTER(_vp)]),
    "amp_disc_destroy": (C.c_int, [_vp]),
    "amp_disc_set_weights": (C.c_int, [_vp, C.POINTER(AmpDiscDesc), _vp]),
    "amp_disc_set_scaler": (C.c_int, [_vp, _vp, _vp, _f32, _f32, _vp]),
    "amp_disc_set_precision": (C.c_int, [_vp, _i32, _vp]),
    "amp_disc_workspace_bytes": (_i64, [_vp, _i64]),
    "amp_disc_input_layout": (C.c_int, [_vp, C.POINTER(AmpDiscInputLayout)]),
    "amp_disc_plan_info": (C.c_int, [_vp, _i64, C.POINTER(AmpDiscPlanInfo)]),
    "amp_disc_set_plan": (C.c_int, [_vp, _i32, _i64]),
    "amp_disc_get_weights": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amp_disc_trainer_create": (C.c_int, [_vp, C.POINTER(AmpDiscTrainCfg), _vp, _vp, C.c_double, _vp, C.POINTER(_vp)]),
    "amp_disc_trainer_destroy": (C.c_int, [_vp]),
    "amp_disc_trainer_scaler": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_double), _vp]),
    "amp_disc_trainer_refresh": (C.c_int, [_vp, _vp]),
    "amp_disc_trainer_adam_state": (C.c_int, [_vp, _vp, _vp, C.POINTER(_i64), _vp]),
    "amp_disc_train_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "amp_converter_create": (C.c_int, [C.POINTER(AmpKinModel), C.POINTER(_vp)]),
    "amp_converter_destroy": (C.c_int, [_vp]),
    "amp_convert_workspace_bytes": (_i64, [_vp, _i64]),
    "amp_convert_motion": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, C.POINTER(AmpConvertOutputs), _vp, _vp]),
    "amp_ring_create": (C.c_int, [_i64, _i32, C.POINTER(_vp)]),
    "amp_ring_destroy": (C.c_int, [_vp]),
    "amp_ring_size": (_i64, [_vp]),
    "amp_ring_head": (_i64, [_vp]),
    "amp_ring_append": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "amp_rows_take_permuted": (C.c_int, [_vp, _i64, _i64, _i32, C.c_uint64, C.c_uint64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "amp_ring_sample": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "amp_disc_style_reward_prescaled": (C.c_int, [_vp, _vp, _i64, _f32, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "amp_disc_style_reward_prescaled_compact": (C.c_int, [_vp, _vp, _i64, _f32, _vp, _f32, _f32, _vp, _vp, _vp, _vp,
                                                          C.POINTER(AmpCompactArgs), _vp]),
    "amp_disc_style_reward_compact": (C.c_int, [_vp, _vp, _i64, _i64, _f32, _vp, _f32, _f32, _vp, _vp, _vp, _vp,
                                                C.POINTER(AmpCompactArgs), _vp]),
    "amp_hot_step": (C.c_int, [C.POINTER(AmpHotStepArgs), _vp]),
    "amp_disc_train_tt_plan": (C.c_int, [_i32, _i32, _i64, _i64, _i64, _i64, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "amp_disc_style_reward": (C.c_int, [_vp, _vp, _i64, _i64, _f32, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle of libamp_engine.so; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP engine has not been built.  Run "
            "`python humanoid_amp_amd/build.py` or `python -c \"import __graft_entry__ as g; g.build()\"` (needs hipcc, "
            "cross-compiles gfx950).  "
            "humanoid_amp_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name):
            raise ImportError(f"{LIB_PATH} does not export {name}: the library is stale, rebuild it with "
                              "`python humanoid_amp_amd/build.py` (humanoid_amp_amd has no CPU fallback)")
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    got = lib.amp_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libamp_engine.so has ABI version {got}, this package expects {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().amp_last_error().decode("utf-8", "replace")
        raise AmpEngineError(f"{what or 'libamp_engine'} failed (code {rc}): {msg}")


def require_gpu(device) -> torch.device:
    """The engine computes on a HIP device only; anything else is an error, never a silent CPU path."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise AmpEngineError(f"humanoid_amp_amd computes on an MI355X (torch device 'cuda:N'); got device '{dev}'. "
                             "There is no CPU fallback.")
    if not torch.cuda.is_available():
        raise AmpEngineError("no HIP device is visible to torch; humanoid_amp_amd has no CPU fallback")
    return dev


def stream_ptr(device=None) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dptr(t, dtype=None, name: str = "tensor") -> C.c_void_p:
    """Device pointer of a contiguous GPU tensor (None -> NULL), with dtype / layout checks."""
    if t is None:
        return C.c_void_p(None)
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise AmpEngineError(f"{name} must be a torch tensor on a HIP device")
    if dtype is not None and t.dtype != dtype:
        raise AmpEngineError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise AmpEngineError(f"{name} must be contiguous")
    return C.c_void_p(t.data_ptr())


def strided_view(t: torch.Tensor, inner: int, name: str):
    """(pointer, env stride in elements) of a [N, inner] fp32 view whose last dim is contiguous."""
    if t.device.type != "cuda" or t.dtype != torch.float32:
        raise AmpEngineError(f"{name} must be a float32 tensor on a HIP device")
    if t.dim() != 2 or t.shape[1] != inner or (inner > 1 and t.stride(1) != 1):
        raise AmpEngineError(f"{name} must be a [N, {inner}] view with a contiguous last dim, got {tuple(t.shape)} / {t.stride()}")
    return C.c_void_p(t.data_ptr()), int(t.stride(0))


class KernelTrace:
    """Context manager around the engine's HIP-event tracer: per-kernel durations of everything launched inside.

    >>> with KernelTrace(capacity=4096, kernel_filter="disc_gemm_kernel<1>") as tr: ...
    >>> tr.summary()   # {name: (calls, total_ms)} -- synchronises the current device first
    """

    def __init__(self, capacity: int = 4096, kernel_filter: str | None = None, every: int = 1):
        """``every`` > 1: bracket only every ``every``-th matching launch (``amp_trace_sample``)."""
        self.capacity, self.filter, self.every = int(capacity), kernel_filter, int(every)

    def __enter__(self):
        check(load().amp_trace_begin(self.capacity, self.filter.encode() if self.filter else None), "amp_trace_begin")
        if self.every > 1:
            check(load().amp_trace_sample(self.every), "amp_trace_sample")
        return self

    def __exit__(self, *exc):
        load().amp_trace_end()
        return False

    def records(self):
        torch.cuda.synchronize()
        lib, out = load(), []
        name, ms = C.create_string_buffer(64), C.c_float()
        for i in range(int(lib.amp_trace_count())):
            check(lib.amp_trace_get(i, name, 64, C.byref(ms)), "amp_trace_get")
            out.append((name.value.decode(), float(ms.value)))
        return out

    def summary(self):
        agg = {}
        for name, ms in self.records():
            c, t = agg.get(name, (0, 0.0))
            agg[name] = (c + 1, t + ms)
        return agg



def calibrate_mfma_f16(random_operands: bool, iters: int = 256, device="cuda:0", reps: int = 3, with_clock: bool = False,
                       layer2_stream: bool = False):
    """TFLOP/s the matrix pipes sustain on a bare fp16 MFMA stream (``amp_calibrate_mfma_f16``): the best of ``reps``
    launches of ~``iters`` x 48 MFMAs per wave, timed by the engine's tracer.  ``with_clock``: also the core clock (MHz)
    the chip sustained inside that launch's MFMA loop (shader-clock ticks / 100 MHz wall ticks, median over workgroups)."""
    lib = load()
    dev = require_gpu(device)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    per_cu = 512 if layer2_stream else 256   # layer2_stream: v_mfma_f32_16x16x32_f16, two waves per SIMD (layer 2's shape)
    mode = int(bool(random_operands)) | (2 if layer2_stream else 0)
    scratch = torch.zeros(cus * per_cu + cus * 4, dtype=torch.float32, device=dev)
    flops = C.c_double()
    best, mhz = 0.0, None
    with torch.cuda.device(dev):
        for _ in range(reps + 1):
            with KernelTrace(capacity=4, kernel_filter="mfma_f16_calibration_kernel") as tr:
                check(lib.amp_calibrate_mfma_f16(mode, int(iters), dptr(scratch), scratch.numel(), C.byref(flops),
                                                 stream_ptr()), "amp_calibrate_mfma_f16")
            ms = tr.records()[-1][1]
            tf = flops.value / (ms * 1e-3) / 1e12
            if tf > best:
                best = tf
                ticks = scratch[cus * per_cu:].view(torch.int64).view(cus, 2).double().cpu()
                mhz = float((ticks[:, 0] / ticks[:, 1].clamp(min=1)).median()) * 100.0
    return (best, mhz) if with_clock else best
