"""CPU: the oracle (our restatement) against the golden vectors produced by running the reference.

Bit-exact everywhere (same ATen ops, same order, same machine class); integer outputs always bit-exact.
"""

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import disc as odisc
from oracle import env as oenv
from oracle import motion as om


def _tables(tag):
    return om.load_tables(gu.clip_files(tag))


@pytest.mark.parametrize("tag", ["g1_walk", "g1_dance", "humanoid3"])
def test_frame_blend_bit_exact(tag):
    fx = gu.golden(f"frame_blend_{tag}")
    mt = _tables(tag)
    assert mt.dt == fx["dt"] and np.array_equal(mt.durations, fx["durations"])
    assert np.array_equal(mt.traj_starts, fx["traj_starts"]) and np.array_equal(mt.traj_ends, fx["traj_ends"])
    i0, i1, b = om.frame_blend(mt, fx["times"], fx["motion_ids"])
    assert np.array_equal(i0, fx["index_0"]) and np.array_equal(i1, fx["index_1"])
    assert np.array_equal(b, fx["blend"])
    # the edge cases the fixture is designed around
    assert (fx["blend"] < -0.5).any() and (fx["blend"] == 0.5).any() and (fx["blend"] == -0.5).any()


@pytest.mark.parametrize("tag", ["g1_walk", "g1_dance", "humanoid3"])
def test_sample_bit_exact(tag):
    fx = gu.golden(f"sample_{tag}")
    mt = _tables(tag)
    outs = om.sample(mt, fx["times"], fx["motion_ids"])
    for name, o in zip(om.TABLE_KEYS, outs):
        assert np.array_equal(o.numpy(), fx[name], equal_nan=True), name
    fx0 = gu.golden(f"sample_defaultids_{tag}")
    outs0 = om.sample(mt, fx0["times"], None)
    for name, o in zip(om.TABLE_KEYS, outs0):
        assert np.array_equal(o.numpy(), fx0[name], equal_nan=True), name


def test_sample_on_frame_returns_table_row():
    """sample(t = k*dt) returns row k of every LERP table exactly."""
    mt = _tables("g1_walk")
    k = np.array([0, 1, 17, 200, 398])
    outs = om.sample(mt, k * mt.dt, np.zeros(5, dtype=np.int64))
    assert torch.equal(outs[0], mt.tables["dof_positions"][k])
    assert torch.equal(outs[2], mt.tables["body_positions"][k])


@pytest.mark.parametrize("name,tag", [("g1_walk_k2", "g1_walk"), ("g1_walk_k10", "g1_walk"),
                                      ("g1_dance_k10", "g1_dance"), ("humanoid3_k2", "humanoid3")])
def test_collect_reference_bit_exact(name, tag):
    fx = gu.golden(f"collect_{name}")
    mt = _tables(tag)
    out = oenv.collect_reference(mt, fx["times"], fx["motion_ids"], int(fx["num_amp_observations"]),
                                 fx["motion_dof_indexes"], int(fx["motion_ref_body_index"]),
                                 fx["motion_key_body_indexes"])
    assert np.array_equal(out.numpy(), fx["amp_obs"])


def test_g1_walk_dof_permutation_matches_survey():
    mt = _tables("g1_walk")
    joints, _ = gu.g1_robot_names()
    perm = [mt.dof_names.index(n) for n in joints]
    assert perm == [0, 6, 12, 1, 7, 13, 2, 8, 14, 3, 9, 15, 22, 4, 10, 16, 23, 5, 11, 17, 24, 18, 25, 19, 26, 20, 27, 21, 28]


def test_reward_functions_bit_exact():
    fx = gu.golden("rewards_fn")
    err = torch.from_numpy(fx["err"])
    assert np.array_equal(oenv.exp_reward_with_floor(err, 1.0, 0.5, 4.0).numpy(), fx["exp_floor_w1_s05_f4"])
    assert np.array_equal(oenv.exp_reward_with_floor(err, 0.7, 0.25, 3.0).numpy(), fx["exp_floor_w07_s025_f3"])
    t = lambda k: torch.from_numpy(fx[k])
    total, log = oenv.compute_rewards(-1.0, -0.1, -10.0, -1.0e-06, -0.001, t("terminated"), t("actions"), t("joint_pos"),
                                      t("soft_joint_pos_limits"), t("joint_acc"), t("joint_vel"))
    assert np.array_equal(total.numpy(), fx["total"])
    for k, v in log.items():
        assert float(v.mean()) == float(fx[k]), k


G1_ENV_CASES = {
    # tag: (clipset, cfg dict)
    "g1_dance_custom": ("g1_dance", dict(K=10, n_actor=1, rew_termination=-1.0, rew_action_l2=-0.1, rew_joint_pos_limits=-10,
                                         rew_joint_acc_l2=-1.0e-06, rew_joint_vel_l2=-0.001, rew_track_vel=1.0)),
    "g1_walk_k2": ("g1_walk", dict(K=2, n_actor=1, rew_termination=-1.0, rew_action_l2=-0.1, rew_joint_pos_limits=-10,
                                   rew_joint_acc_l2=-1.0e-06, rew_joint_vel_l2=-0.001, rew_track_vel=1.0)),
    "g1_deploy_hist3": ("g1_walk", dict(K=2, n_actor=3, rew_termination=-1.0, rew_action_l2=-0.1, rew_joint_pos_limits=0.0,
                                        rew_joint_acc_l2=0.0, rew_joint_vel_l2=0.0, rew_track_vel=1.0)),
    "g1_deploy_hist2_ablate": ("g1_walk", dict(K=2, n_actor=2, hist_actions=False, hist_command=False, rew_termination=0.0,
                                               rew_action_l2=0.0, rew_joint_pos_limits=0.0, rew_joint_acc_l2=0.0,
                                               rew_joint_vel_l2=0.0, rew_track_vel=1.0)),
}


@pytest.mark.parametrize("tag", list(G1_ENV_CASES))
def test_g1_env_step_sequence(tag):
    clipset, cfg = G1_ENV_CASES[tag]
    fx = gu.golden(f"envstep_{tag}")
    mt = _tables(clipset)
    ref, keys = int(fx["ref_body_index"]), fx["key_body_indexes"]
    perm = fx["motion_dof_indexes"]
    m_ref = mt.body_names.index("pelvis")
    m_keys = [mt.body_names.index(n) for n in gu.G1_KEY_BODIES]
    K = cfg["K"]
    amp = fx["init_amp_observation_buffer"].copy()
    hist = torch.from_numpy(fx["init_actor_obs_history_buffer"].copy()) if cfg["n_actor"] > 1 else None
    N = amp.shape[0]
    lim = torch.from_numpy(np.broadcast_to(fx["soft_joint_pos_limits"], (N,) + fx["soft_joint_pos_limits"].shape[1:]).copy())
    for s in range(int(fx["n_steps"])):
        p = f"s{s}_"
        t = lambda k: torch.from_numpy(fx[p + k])
        died, time_out = oenv.dones(t("in_episode_length_buf"), int(fx["max_episode_length"]),
                                    t("in_body_pos_w")[:, ref, 2], 0.5)
        assert np.array_equal(died.numpy(), fx[p + "out_died"]) and np.array_equal(time_out.numpy(), fx[p + "out_time_out"])
        total, parts = oenv.g1_task_reward(cfg, t("in_body_lin_vel_w")[:, ref], t("in_body_quat_w")[:, ref],
                                           t("in_command_target_speed"), died, t("in_actions"), t("in_joint_pos"), lim,
                                           t("in_joint_acc"), t("in_joint_vel"))
        assert np.array_equal(total.numpy(), fx[p + "out_reward"])
        for k, v in parts.items():
            assert float(v.mean()) == float(fx[p + "log_" + k]), k
        ids = oenv.reset_env_ids(died, time_out)
        assert ids.dtype == torch.int64 and np.array_equal(ids.numpy(), fx[p + "out_reset_env_ids"])
        if len(ids):
            rt, rid = fx[p + "reset_times"], fx[p + "reset_motion_ids"]
            root, dpos, dvel = oenv.reset_reference_state(mt, rt, rid, perm, m_ref, torch.from_numpy(fx["env_origins"])[ids], 0.05)
            # default_root_state is zeros in the fixture, so root == out_reset_root_state
            assert np.array_equal(root.numpy(), fx[p + "out_reset_root_state"])
            assert np.array_equal(dpos.numpy(), fx[p + "out_reset_dof_pos"])
            assert np.array_equal(dvel.numpy(), fx[p + "out_reset_dof_vel"])
            rows = oenv.collect_reference(mt, rt, rid, K, perm, m_ref, m_keys).view(len(ids), K, -1)
            assert np.array_equal(rows.numpy(), fx[p + "out_reset_amp_rows"])
        st, amp_in = gu.obs_inputs(fx, s, amp)
        tt = lambda k: torch.from_numpy(st[k])
        obs = oenv.compute_obs(tt("joint_pos"), tt("joint_vel"), tt("body_pos_w")[:, ref], tt("body_quat_w")[:, ref],
                               tt("body_lin_vel_w")[:, ref], tt("body_ang_vel_w")[:, ref], tt("body_pos_w")[:, keys])
        buf = torch.from_numpy(amp_in)
        amp_obs = oenv.shift_history(buf, obs)
        assert np.array_equal(amp_obs.numpy(), fx[p + "out_amp_obs"])
        jr = torch.from_numpy(fx[p + "obsin_just_reset_mask"].copy()) if cfg["n_actor"] > 1 else None
        pol = oenv.actor_observation(obs, t("obsin_last_actions"), t("obsin_command_target_speed"), use_command=True,
                                     n_actor=cfg["n_actor"], hist_buf=hist, just_reset=jr,
                                     hist_actions=cfg.get("hist_actions", True), hist_command=cfg.get("hist_command", True))
        assert np.array_equal(pol.numpy(), fx[p + "out_policy_obs"])
        if hist is not None:
            assert np.array_equal(hist.numpy(), fx[p + "out_actor_obs_history_buffer"])
        amp = buf.numpy().copy()
    if tag == "g1_deploy_hist3":
        assert fx["s0_out_policy_obs"].shape[1] == int(gu.golden("meta")["deploy_hist3_observation_space"])


def test_humanoid_env_step_sequence():
    fx = gu.golden("envstep_humanoid3")
    ref, keys = int(fx["ref_body_index"]), fx["key_body_indexes"]
    amp = fx["init_amp_observation_buffer"].copy()
    for s in range(2):
        p = f"s{s}_"
        t = lambda k: torch.from_numpy(fx[p + k])
        died, time_out = oenv.dones(t("in_episode_length_buf"), int(fx["max_episode_length"]), t("in_body_pos_w")[:, ref, 2], 0.5)
        assert np.array_equal(died.numpy(), fx[p + "out_died"]) and np.array_equal(time_out.numpy(), fx[p + "out_time_out"])
        assert np.array_equal(oenv.reset_env_ids(died, time_out).numpy(), fx[p + "out_reset_env_ids"])
        obs = oenv.compute_obs(t("in_joint_pos"), t("in_joint_vel"), t("in_body_pos_w")[:, ref], t("in_body_quat_w")[:, ref],
                               t("in_body_lin_vel_w")[:, ref], t("in_body_ang_vel_w")[:, ref], t("in_body_pos_w")[:, keys])
        buf = torch.from_numpy(amp)
        assert np.array_equal(oenv.shift_history(buf, obs).numpy(), fx[p + "out_amp_obs"])
        assert np.array_equal(obs.numpy(), fx[p + "out_policy_obs"])  # humanoid policy obs == AMP frame
        amp = buf.numpy().copy()


def test_disc_oracle_regression():
    """UNPINNED by the reference (skrl absent): regression pin of our own restatement + fp64 arbitration."""
    fx = gu.golden("disc_k2_166")
    w = odisc.make_weights(166, seed=int(fx["seed"]))
    assert np.array_equal(w[0][0][0, :8].numpy(), fx["w1_head"]) and np.array_equal(w[2][0].numpy(), fx["w3"])
    out = odisc.forward(w, torch.from_numpy(fx["amp_obs"]), torch.from_numpy(fx["running_mean"]),
                        torch.from_numpy(fx["running_variance"]), task=torch.from_numpy(fx["task_reward"]),
                        task_w=0.5, style_w=0.5)
    assert np.array_equal(out["scaled"].numpy(), fx["scaled"])
    np.testing.assert_allclose(out["logits"].numpy(), fx["logits"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["logits"].numpy(), fx["logits_f64"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["style"].numpy(), fx["style_reward"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(out["combined"].numpy(), fx["combined"], rtol=0, atol=5e-6)
