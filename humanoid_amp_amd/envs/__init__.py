"""Task environments (reference: g1_amp_env.py, humanoid_amp_env.py) and their configs."""

from .amp_env import G1AmpEnv, HumanoidAmpEnv
from .cfg import (TASKS, G1AmpCustomEnvCfg, G1AmpDanceEnvCfg, G1AmpDeployEnvCfg, G1AmpEnvCfg, G1AmpEnvCfg_CUSTOM,
                  G1AmpWalkEnvCfg, HumanoidAmpDanceEnvCfg, HumanoidAmpEnvCfg, HumanoidAmpRunEnvCfg, HumanoidAmpWalkEnvCfg)
from .direct_rl_env import DirectRLEnv
from .sim import SyntheticArticulation


def make(task_id: str, num_envs: int | None = None, device: str | None = None, **kwargs):
    """``gym.make`` counterpart for the reference's seven task ids (reference __init__.py:18-93)."""
    cls_name, cfg_cls, _ = TASKS[task_id]
    cfg = cfg_cls()
    if num_envs is not None:
        cfg.scene.num_envs = int(num_envs)
    if device is not None:
        cfg.sim.device = device
    try:
        return {"G1AmpEnv": G1AmpEnv, "HumanoidAmpEnv": HumanoidAmpEnv}[cls_name](cfg, **kwargs)
    except (ValueError, FileNotFoundError) as e:
        if "No files found" in str(e) or isinstance(e, FileNotFoundError):
            # same exception type as the reference's resolver (motion_loader.py:55,84), with the way out spelled out
            raise ValueError(f"{task_id}: motion clip(s) {cfg.motion_file!r} are not shipped with this package "
                             "(the reference does not distribute them either); pass your own via cfg.motion_file -- "
                             f"{e}") from e
        raise


__all__ = ["G1AmpEnv", "HumanoidAmpEnv", "DirectRLEnv", "SyntheticArticulation", "TASKS", "make", "G1AmpEnvCfg",
           "G1AmpEnvCfg_CUSTOM", "G1AmpWalkEnvCfg", "G1AmpDanceEnvCfg", "G1AmpCustomEnvCfg", "G1AmpDeployEnvCfg",
           "HumanoidAmpEnvCfg", "HumanoidAmpDanceEnvCfg", "HumanoidAmpRunEnvCfg", "HumanoidAmpWalkEnvCfg"]
