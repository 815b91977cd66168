"""Stand-in modules for the third-party packages the reference env files import.

Used ONLY by ``gen_golden.py`` (this container, where ``/root/reference`` is mounted) so
that the reference's own ``g1_amp_env.py`` / ``humanoid_amp_env.py`` can be imported by
file path and their functions executed to produce golden vectors.  Nothing here is part
of the product, and nothing here travels as "the reference".

Isaac Lab / gymnasium / skrl are NOT installed in this image.  Two functions of
``isaaclab.utils.math`` are called by the reference on the hot path
(g1_amp_env.py:16,253,495-496; humanoid_amp_env.py:16,257-258).  They are restated below
from the published Isaac Lab 2.2.0 formulae.  Their fp32 operation
order is therefore *our* restatement -> every golden value that flows through them is
"parity unpinned" w.r.t. Isaac Lab itself (the mathematics was cross-checked against
scipy's Rotation in the survey).

The file must be a real ``.py`` file: TorchScript compiles the reference's
``@torch.jit.script`` functions from source and resolves ``quat_apply`` through the
importing module's globals.
"""

from __future__ import annotations

import sys
import types

import torch


# --- restated third-party math (Isaac Lab 2.2.0, isaaclab/utils/math.py) ------------------------


@torch.jit.script
def quat_apply(quat: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    """Rotate ``vec`` by ``quat`` (wxyz):  v + w*t + q_xyz x t,  t = 2*(q_xyz x v)."""
    shape = vec.shape
    quat = quat.reshape(-1, 4)
    vec = vec.reshape(-1, 3)
    xyz = quat[:, 1:]
    t = xyz.cross(vec, dim=-1) * 2
    return (vec + quat[:, 0:1] * t + xyz.cross(t, dim=-1)).view(shape)


@torch.jit.script
def quat_rotate_inverse(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Rotate ``v`` by the inverse of ``q`` (wxyz):  v(2w^2-1) - 2w(q_v x v) + 2 q_v (q_v . v)."""
    q_w = q[..., 0]
    q_vec = q[..., 1:]
    a = v * (2.0 * q_w**2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    if q_vec.dim() == 2:
        c = q_vec * torch.bmm(q_vec.view(q.shape[0], 1, 3), v.view(q.shape[0], 3, 1)).squeeze(-1) * 2.0
    else:
        c = q_vec * torch.einsum("...i,...i->...", q_vec, v).unsqueeze(-1) * 2.0
    return a - b + c


# --- inert placeholders ---------------------------------------------------------------------------


class _Cfg:
    """Accepts any constructor arguments; ``replace`` returns self; attribute access is lazy."""

    def __init__(self, *args, **kwargs):
        self.__dict__.update(kwargs)

    def replace(self, **kwargs):
        return self

    class InitialStateCfg:
        def __init__(self, *args, **kwargs):
            self.__dict__.update(kwargs)


class _AnyAttrModule(types.ModuleType):
    """A module whose every unknown attribute is the inert ``_Cfg`` class."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Cfg


class _Box:
    def __init__(self, low=None, high=None, shape=None, **kwargs):
        self.low, self.high, self.shape = low, high, shape


def _module(name: str, cls=types.ModuleType, **attrs) -> types.ModuleType:
    mod = cls(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def install() -> None:
    """Register the stand-in packages in ``sys.modules`` (idempotent)."""
    if "isaaclab" in sys.modules and getattr(sys.modules["isaaclab"], "_amp_stub", False):
        return
    pkg = _module("isaaclab", _amp_stub=True)
    pkg.__path__ = []
    sim = _module("isaaclab.sim", _AnyAttrModule)
    sim.__path__ = []
    spawners = _module("isaaclab.sim.spawners", _AnyAttrModule)
    spawners.__path__ = []
    _module("isaaclab.sim.spawners.from_files", _AnyAttrModule, spawn_ground_plane=lambda *a, **k: None)
    _module("isaaclab.assets", Articulation=_Cfg, ArticulationCfg=_Cfg)
    _module("isaaclab.actuators", _AnyAttrModule)
    _module("isaaclab.envs", DirectRLEnv=object, DirectRLEnvCfg=object)
    _module("isaaclab.scene", _AnyAttrModule)
    utils = _module("isaaclab.utils", configclass=lambda c: c)
    utils.__path__ = []
    _module("isaaclab.utils.math", quat_apply=quat_apply, quat_rotate_inverse=quat_rotate_inverse)
    _module("isaaclab_assets", HUMANOID_28_CFG=_Cfg())
    spaces = _module("gymnasium.spaces", Box=_Box)
    gym = _module("gymnasium", spaces=spaces, register=lambda *a, **k: None)
    gym.__path__ = []
