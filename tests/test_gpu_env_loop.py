"""GPU: the drop-in env classes stepped end to end (synthetic articulation) against an oracle shadow.

Every hook's inputs are snapshotted when the hook runs and replayed through the oracle; the oracle also keeps its
own AMP history buffer across steps, so drift would show.  Bars as in test_gpu_env.py.
"""

import numpy as np
import pytest
import torch

from oracle import env as oenv
from oracle import motion as om

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _shadow_run(env, mt, keys_names, root_body, z_lift, steps, is_g1):
    cfg = env.cfg
    K, N = cfg.num_amp_observations, env.num_envs
    perm, ref = env.motion_dof_indexes, env.motion_ref_body_index
    m_keys = [mt.body_names.index(n) for n in keys_names]
    snap = {}
    orig_dones, orig_reset = env._get_dones, env._reset_strategy_random

    def dones_hook():
        d = env.robot.data
        snap["pre"] = {k: getattr(d, k).clone().cpu() for k in ("joint_pos", "joint_vel", "joint_acc", "body_pos_w",
                                                                "body_quat_w", "body_lin_vel_w")}
        snap["ep"] = env.episode_length_buf.clone().cpu()
        snap["cmd"] = env.command_target_speed.clone().cpu()
        snap["act"] = env.actions.clone().cpu()
        return orig_dones()

    def reset_hook(env_ids, start=False):
        state = np.random.get_state()
        ids, times = env._motion_loader.sample_times(env_ids.shape[0], start=start)
        np.random.set_state(state)
        snap["reset"] = (env_ids.clone().cpu(), np.asarray(ids), np.asarray(times))
        return orig_reset(env_ids, start)

    env._get_dones, env._reset_strategy_random = dones_hook, reset_hook
    np.random.seed(11)
    torch.manual_seed(11)
    obs, _ = env.reset()
    r = env.ref_body_index
    shadow = env.amp_observation_buffer.clone().cpu()  # after reset + first obs: adopt, then track independently
    n_actor = getattr(cfg, "num_actor_observations", 1)
    if n_actor > 1:  # actor history (g1_amp_env.py:198-235): adopt after the warm start, then track independently
        shadow_hist = env.actor_obs_history_buffer.clone().cpu()
        shadow_just = torch.zeros(N, dtype=torch.bool)
        assert not bool(env._just_reset_mask.any())
    n_resets = 0
    for step in range(steps):
        snap.pop("reset", None)
        a = torch.randn(N, cfg.action_space, device=env.device) * 0.3
        last_before = env.last_actions.clone().cpu() if is_g1 else None
        obs, rew, term, tout, extras = env.step(a)
        pre = snap["pre"]
        died, time_out = oenv.dones(snap["ep"], env.max_episode_length, pre["body_pos_w"][:, r, 2], cfg.termination_height)
        assert torch.equal(term.cpu(), died) and torch.equal(tout.cpu(), time_out)
        want_ids = oenv.reset_env_ids(died, time_out)
        if is_g1:
            rc = {k: float(getattr(cfg, k)) for k in ("rew_termination", "rew_action_l2", "rew_joint_pos_limits",
                                                      "rew_joint_acc_l2", "rew_joint_vel_l2", "rew_track_vel")}
            total, parts = oenv.g1_task_reward(rc, pre["body_lin_vel_w"][:, r], pre["body_quat_w"][:, r], snap["cmd"], died,
                                               snap["act"], pre["joint_pos"], env.robot.data.soft_joint_pos_limits.cpu(),
                                               pre["joint_acc"], pre["joint_vel"])
            scale = max(1.0, float(total.abs().max()))
            assert float((rew.cpu() - total).abs().max()) <= TOL * scale
            for k, v in parts.items():
                assert abs(extras["log"][k] - float(v.mean())) <= 2e-5 * max(1.0, abs(float(v.mean()))), k
        else:
            assert float((rew - 1.0).abs().max()) == 0.0
        if len(want_ids):
            ids, m_ids, m_t = snap["reset"]
            assert torch.equal(ids, want_ids)  # bit-exact ascending ids
            n_resets += len(ids)
            rows = oenv.collect_reference(mt, m_t, m_ids, K, perm, ref, m_keys).view(len(ids), K, -1)
            shadow[ids] = rows
            if n_actor > 1:
                shadow_just[ids] = True  # _reset_idx: _just_reset_mask[env_ids] = True (g1_amp_env.py:352-358)
            root, dpos, dvel = oenv.reset_reference_state(mt, m_t, m_ids, perm, mt.body_names.index(root_body),
                                                          env.scene.env_origins.cpu()[ids], z_lift)
            d = env.robot.data
            assert float((d.joint_pos[ids.to(env.device)].cpu() - dpos).abs().max()) == 0.0
            assert float((d.body_pos_w[ids.to(env.device), r].cpu() - root[:, :3]).abs().max()) == 0.0
            assert float((d.body_quat_w[ids.to(env.device), r].cpu() - root[:, 3:7]).abs().max()) <= TOL
        d = env.robot.data
        ob = oenv.compute_obs(d.joint_pos.cpu(), d.joint_vel.cpu(), d.body_pos_w[:, r].cpu(), d.body_quat_w[:, r].cpu(),
                              d.body_lin_vel_w[:, r].cpu(), d.body_ang_vel_w[:, r].cpu(),
                              d.body_pos_w[:, env.key_body_indexes].cpu())
        amp = oenv.shift_history(shadow, ob)
        got = extras["amp_obs"]
        assert got.data_ptr() == env.amp_observation_buffer.data_ptr()  # a view, as in the reference
        assert float((got.cpu() - amp).abs().max()) <= TOL
        if is_g1:
            la = env.last_actions.clone().cpu()
            if n_actor > 1:
                pol = oenv.actor_observation(ob, la, env.command_target_speed.cpu(), use_command=cfg.rew_track_vel > 0.0, n_actor=n_actor,
                                             hist_buf=shadow_hist, just_reset=shadow_just,
                                             hist_actions=cfg.history_include_last_actions, hist_command=cfg.history_include_command)
                assert obs["policy"].shape[1] == cfg.observation_space
                assert float((env.actor_obs_history_buffer.cpu() - shadow_hist).abs().max()) <= TOL
                assert not bool(env._just_reset_mask.any())
            else:
                pol = oenv.actor_observation(ob, la, env.command_target_speed.cpu(), use_command=cfg.rew_track_vel > 0.0)
            assert float((obs["policy"].cpu() - pol).abs().max()) <= TOL
        else:
            assert float((obs["policy"].cpu() - ob).abs().max()) <= TOL
    return n_resets


def test_g1_dance_env_loop():
    from humanoid_amp_amd.envs import G1AmpEnv, G1AmpDanceEnvCfg
    from humanoid_amp_amd.robots import G1_KEY_BODY_NAMES

    cfg = G1AmpDanceEnvCfg()
    cfg.scene.num_envs = 200
    cfg.episode_length_s = 0.2  # 12 steps: forces time-outs so the reset path runs
    env = G1AmpEnv(cfg)
    assert env.amp_observation_size == 830 and env.amp_observation_space.shape == (830,)
    mt = om.load_tables([cfg.motion_file])
    n = _shadow_run(env, mt, G1_KEY_BODY_NAMES, "pelvis", 0.05, steps=30, is_g1=True)
    assert n > 200
    out = env.collect_reference_motions(16)
    assert out.shape == (16, 830)


def test_g1_walk_env_loop_random_reset():
    from humanoid_amp_amd.envs import G1AmpEnv, G1AmpEnvCfg_CUSTOM, G1AmpWalkEnvCfg
    from humanoid_amp_amd.robots import G1_KEY_BODY_NAMES

    cfg = G1AmpEnvCfg_CUSTOM(motion_file=G1AmpWalkEnvCfg().motion_file, num_amp_observations=2, reset_strategy="random")
    cfg.scene.num_envs = 130
    cfg.episode_length_s = 0.15
    env = G1AmpEnv(cfg)
    mt = om.load_tables([cfg.motion_file])
    assert _shadow_run(env, mt, G1_KEY_BODY_NAMES, "pelvis", 0.05, steps=25, is_g1=True) > 100


@pytest.mark.parametrize("n_actor,hist_actions,hist_command", [(2, True, True), (3, False, True)])
def test_g1_deploy_env_loop_actor_history(n_actor, hist_actions, hist_command):
    """The Deploy task (g1_amp_env_cfg.py:160-206): multi-clip yaml table, K = 10, actor history with reset warm start -- the
    history buffer and the policy row follow an independent oracle shadow over 25 steps with resets."""
    from humanoid_amp_amd.envs import G1AmpDeployEnvCfg, G1AmpEnv
    from humanoid_amp_amd.motions.motion_loader import _resolve_motion_files
    from humanoid_amp_amd.robots import G1_KEY_BODY_NAMES

    cfg = G1AmpDeployEnvCfg(num_actor_observations=n_actor, history_include_last_actions=hist_actions,
                            history_include_command=hist_command)
    cfg.scene.num_envs = 150
    cfg.episode_length_s = 0.2
    env = G1AmpEnv(cfg)
    mt = om.load_tables(_resolve_motion_files(cfg.motion_file))
    assert _shadow_run(env, mt, G1_KEY_BODY_NAMES, "pelvis", 0.05, steps=25, is_g1=True) > 150


def test_humanoid_env_loop():
    from humanoid_amp_amd.envs import HumanoidAmpEnv, HumanoidAmpWalkEnvCfg
    from humanoid_amp_amd.robots import HUMANOID_KEY_BODY_NAMES

    cfg = HumanoidAmpWalkEnvCfg()
    cfg.scene.num_envs = 96
    cfg.episode_length_s = 0.4
    env = HumanoidAmpEnv(cfg)
    mt = om.load_tables([cfg.motion_file])
    assert _shadow_run(env, mt, HUMANOID_KEY_BODY_NAMES, "torso", 0.15, steps=30, is_g1=False) > 96


def test_make_and_errors():
    from humanoid_amp_amd.envs import G1AmpEnv, G1AmpWalkEnvCfg, make

    env = make("Isaac-G1-AMP-Walk-Direct-v0", num_envs=64)
    assert isinstance(env, G1AmpEnv) and env.cfg.observation_space == 100
    obs, extras = env.reset(seed=0)
    assert obs["policy"].shape == (64, 100) and extras["amp_obs"].shape == (64, 166)
    bad = G1AmpWalkEnvCfg(reset_strategy="bogus")
    bad.scene.num_envs = 8
    with pytest.raises(ValueError, match="Unknown reset strategy"):
        G1AmpEnv(bad).reset()
    # the Custom task runs on the reference's own custom_motion.npz (309 frames, 25 bodies; shipped as data since round 3)
    cus = make("Isaac-G1-AMP-Custom-Direct-v0", num_envs=48)
    assert cus.cfg.num_amp_observations == 10 and cus.amp_observation_size == 830
    assert cus.motion_key_body_indexes == [19, 20, 21, 22] and cus.motion_ref_body_index == 0
    obs, extras = cus.reset(seed=0)
    obs, rew, term, tout, extras = cus.step(torch.zeros(48, 29, device="cuda"))
    assert extras["amp_obs"].shape == (48, 830) and torch.isfinite(extras["amp_obs"]).all() and torch.isfinite(rew).all()
    # a clip that does not exist: the reference's resolver error type (motion_loader.py:55,84)
    missing = G1AmpWalkEnvCfg(motion_file="/nonexistent/clip.npz")
    missing.scene.num_envs = 8
    with pytest.raises((ValueError, FileNotFoundError)):
        G1AmpEnv(missing)
    # the Deploy task runs on the bundled motion_config.yaml (the reference's own list points at private recordings)
    dep = make("Isaac-G1-AMP-Deploy-Direct-v0", num_envs=32)
    obs, extras = dep.reset(seed=0)
    assert obs["policy"].shape == (32, dep.cfg.observation_space) and extras["amp_obs"].shape == (32, dep.amp_observation_size)
    obs, rew, term, tout, extras = dep.step(torch.zeros(32, 29, device="cuda"))
    assert torch.isfinite(obs["policy"]).all() and torch.isfinite(rew).all()
    # fixed command of the Deploy cfg (track_vel_range lo == hi): reset writes (lo, 0) and an infinite timer
    assert torch.equal(dep.command_target_speed, torch.tensor([[1.0, 0.0]], device="cuda").expand(32, 2))
    assert torch.isinf(dep.command_time_left).all()
