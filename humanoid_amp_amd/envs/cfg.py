"""Plain-dataclass counterparts of the reference's config classes (values only).

Reference: g1_amp_env_cfg.py:22-206, humanoid_amp_env_cfg.py:23-93.  The Isaac Lab ``configclass`` /
``SimulationCfg`` / ``ArticulationCfg`` plumbing needs Isaac Sim and is out of scope; ``sim`` and ``scene`` keep
only the fields the hot path reads (dt, device, num_envs, env_spacing).

Deviation, on purpose: the reference's base ``G1AmpEnvCfg`` is stale (amp_observation_space = 101 and
observation_space = 71 cannot work with the 83-float frames / 100-float actor observation its env emits, SURVEY.md
section 0.1); here it carries the self-consistent values 83 / 100.
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field

MOTIONS_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "motions")


@dataclass
class SimCfg:
    dt: float = 1 / 60
    device: str = "cuda:0"


@dataclass
class SceneCfg:
    num_envs: int = 4096
    env_spacing: float = 4.0


@dataclass
class G1AmpEnvCfg:
    # reward (g1_amp_env_cfg.py:26-32)
    rew_termination: float = -0
    rew_action_l2: float = -0.00
    rew_joint_pos_limits: float = -0
    rew_joint_acc_l2: float = -0.00
    rew_joint_vel_l2: float = -0.00
    rew_track_vel: float = 0.0
    # env
    episode_length_s: float = 10.0
    decimation: int = 2
    track_vel_range: tuple = (0.0, 0.0)
    command_resampling_time_range: tuple = (4.0, 7.0)
    # spaces
    observation_space: int = 100   # 71 base + 29 last actions (reference value 71 is stale)
    action_space: int = 29
    state_space: int = 0
    num_amp_observations: int = 2
    amp_observation_space: int = 83  # reference value 71 + 3*10 is stale
    num_actor_observations: int = 1
    early_termination: bool = True
    termination_height: float = 0.5
    motion_file: str = ""
    reference_body: str = "pelvis"
    reset_strategy: str = "random"  # default | random | random-start
    sim: SimCfg = field(default_factory=SimCfg)
    scene: SceneCfg = field(default_factory=SceneCfg)


@dataclass
class G1AmpEnvCfg_CUSTOM(G1AmpEnvCfg):
    # g1_amp_env_cfg.py:81-141
    rew_termination: float = -1.0
    rew_action_l2: float = -0.1
    rew_joint_pos_limits: float = -10
    rew_joint_acc_l2: float = -1.0e-06
    rew_joint_vel_l2: float = -0.001
    rew_track_vel: float = 1.0
    decimation: int = 1
    track_vel_range: tuple = (-1.0, 1.0)
    observation_space: int = 102
    num_amp_observations: int = 10
    reset_strategy: str = "random-start"


@dataclass
class G1AmpWalkEnvCfg(G1AmpEnvCfg):
    motion_file: str = os.path.join(MOTIONS_DIR, "G1_walk.npz")


@dataclass
class G1AmpDanceEnvCfg(G1AmpEnvCfg_CUSTOM):
    motion_file: str = os.path.join(MOTIONS_DIR, "G1_dance.npz")


@dataclass
class G1AmpCustomEnvCfg(G1AmpEnvCfg_CUSTOM):
    episode_length_s: float = 5.0
    motion_file: str = os.path.join(MOTIONS_DIR, "custom_motion.npz")  # the reference's own clip, shipped as data


@dataclass
class G1AmpDeployEnvCfg(G1AmpEnvCfg_CUSTOM):
    # g1_amp_env_cfg.py:160-206
    episode_length_s: float = 10.0
    motion_file: str = os.path.join(MOTIONS_DIR, "motion_config.yaml")
    reset_strategy: str = "random"
    track_vel_range: tuple = (1.0, 1.0)
    rew_termination: float = 0.0
    rew_action_l2: float = 0.0
    rew_joint_pos_limits: float = 0.0
    rew_joint_acc_l2: float = 0.0
    rew_joint_vel_l2: float = 0.0
    rew_track_vel: float = 1.0
    num_actor_observations: int = 2
    history_include_last_actions: bool = True
    history_include_command: bool = True

    def __post_init__(self):
        base = self.amp_observation_space - 4 * 3
        cmd = 2 if self.rew_track_vel > 0.0 else 0
        cur = base + self.action_space + cmd
        if self.num_actor_observations <= 1:
            self.observation_space = cur
        else:
            hist = base + (self.action_space if self.history_include_last_actions else 0) + \
                (cmd if self.history_include_command else 0)
            self.observation_space = cur + (self.num_actor_observations - 1) * hist


@dataclass
class HumanoidAmpEnvCfg:
    # humanoid_amp_env_cfg.py:23-78
    episode_length_s: float = 10.0
    decimation: int = 2
    observation_space: int = 81
    action_space: int = 28
    state_space: int = 0
    num_amp_observations: int = 2
    amp_observation_space: int = 81
    early_termination: bool = True
    termination_height: float = 0.5
    motion_file: str = ""
    reference_body: str = "torso"
    reset_strategy: str = "random"
    sim: SimCfg = field(default_factory=SimCfg)
    scene: SceneCfg = field(default_factory=lambda: SceneCfg(env_spacing=10.0))


@dataclass
class HumanoidAmpDanceEnvCfg(HumanoidAmpEnvCfg):
    motion_file: str = os.path.join(MOTIONS_DIR, "humanoid_dance.npz")


@dataclass
class HumanoidAmpRunEnvCfg(HumanoidAmpEnvCfg):
    motion_file: str = os.path.join(MOTIONS_DIR, "humanoid_run.npz")


@dataclass
class HumanoidAmpWalkEnvCfg(HumanoidAmpEnvCfg):
    motion_file: str = os.path.join(MOTIONS_DIR, "humanoid_walk.npz")


# gym task id -> (env class name, cfg class, skrl yaml of the reference) -- __init__.py:18-93 of the reference, as data
TASKS = {
    "Isaac-Humanoid-AMP-Dance-Direct-v0": ("HumanoidAmpEnv", HumanoidAmpDanceEnvCfg, "skrl_dance_amp_cfg.yaml"),
    "Isaac-Humanoid-AMP-Run-Direct-v0": ("HumanoidAmpEnv", HumanoidAmpRunEnvCfg, "skrl_run_amp_cfg.yaml"),
    "Isaac-Humanoid-AMP-Walk-Direct-v0": ("HumanoidAmpEnv", HumanoidAmpWalkEnvCfg, "skrl_walk_amp_cfg.yaml"),
    "Isaac-G1-AMP-Walk-Direct-v0": ("G1AmpEnv", G1AmpWalkEnvCfg, "skrl_g1_walk_amp_cfg.yaml"),
    "Isaac-G1-AMP-Dance-Direct-v0": ("G1AmpEnv", G1AmpDanceEnvCfg, "skrl_g1_dance_amp_cfg.yaml"),
    "Isaac-G1-AMP-Custom-Direct-v0": ("G1AmpEnv", G1AmpCustomEnvCfg, "skrl_g1_custom_amp_cfg.yaml"),
    "Isaac-G1-AMP-Deploy-Direct-v0": ("G1AmpEnv", G1AmpDeployEnvCfg, "skrl_g1_deploy_amp_cfg.yaml"),
}
