"""humanoid_amp_amd -- MI355X-native AMP observation / motion-sample / reward engine.

Drop-in for the hot path of zhoushanghai/humanoid_amp (MotionLoader.sample, compute_obs + AMP history,
dones + reset-id compaction, task reward, discriminator style reward) behind the reference's own class
surfaces.  All numerics run in ``csrc/libamp_engine.so`` (hand-written HIP for gfx950, C ABI in
``include/amp_engine.h``); importing this package fails loudly if that library is not built.
"""

from . import _native

_native.load()  # no CPU fallback: a missing / stale library is an ImportError here

from .engine import AmpDiscriminator, EnvStepConfig, EnvStepKernel, reset_compact  # noqa: E402
from .envs import G1AmpEnv, HumanoidAmpEnv, make  # noqa: E402
from .motions import MotionLoader  # noqa: E402

__all__ = ["MotionLoader", "G1AmpEnv", "HumanoidAmpEnv", "make", "AmpDiscriminator", "EnvStepConfig", "EnvStepKernel",
           "reset_compact"]
