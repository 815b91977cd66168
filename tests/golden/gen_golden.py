#!/usr/bin/env python3
"""Generate the golden vectors under ``tests/golden/`` by RUNNING THE REFERENCE.

Runs only where ``/root/reference`` is mounted (the build container).  The reference's own
``motions/motion_loader.py``, ``g1_amp_env.py`` and ``humanoid_amp_env.py`` are imported by file
path (third-party Isaac Lab / gymnasium imports satisfied by ``_isaac_stubs``) and their functions /
unbound methods are executed on seeded synthetic inputs.  Only arrays (inputs + the reference's
outputs) are written, as ``.npz`` files.  No reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

What each fixture pins (reference file:line):
  frame_blend_*.npz   MotionLoader._compute_frame_blend        motions/motion_loader.py:281-307
  sample_*.npz        MotionLoader.sample (6-tuple)            motions/motion_loader.py:331-390
  collect_*.npz       G1AmpEnv.collect_reference_motions       g1_amp_env.py:445-486
                      (HumanoidAmpEnv.collect_reference_motions humanoid_amp_env.py:219-248 raises in this
                       fork for K > 1 -- see main(); humanoid clips go through the G1 convention)
  envstep_*.npz       G1AmpEnv._get_dones/_get_rewards/_reset_strategy_random/_get_observations
                                                                g1_amp_env.py:175-242,246-330,371-441
                      HumanoidAmpEnv._get_observations/_get_dones humanoid_amp_env.py:105-140
  rewards_fn.npz      exp_reward_with_floor / compute_rewards   g1_amp_env.py:500-532,564-606
  disc_*.npz          NOT from the reference (skrl is absent): restated scaler + MLP + style reward,
                      "parity unpinned" (SURVEY.md §8c).

Values that pass through ``quat_apply`` / ``quat_rotate_inverse`` use the restated Isaac Lab math of
``_isaac_stubs.py`` (third-party, absent) -> unpinned fp32 op order for those columns only.
"""

from __future__ import annotations

import importlib.util
import math
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = HERE

G1_KEY_BODIES = ["right_rubber_hand", "left_rubber_hand", "right_ankle_roll_link", "left_ankle_roll_link"]
HUM_KEY_BODIES = ["right_hand", "left_hand", "right_foot", "left_foot"]


def _load(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("gen_golden.py needs /root/reference (build container only); refusing to run.")
    sys.path.insert(0, HERE)
    import _isaac_stubs

    _isaac_stubs.install()
    pkg = types.ModuleType("humanoid_amp")
    pkg.__path__ = [REF]
    sys.modules["humanoid_amp"] = pkg
    mpkg = types.ModuleType("humanoid_amp.motions")
    mpkg.__path__ = [os.path.join(REF, "motions")]
    sys.modules["humanoid_amp.motions"] = mpkg
    ml = _load("humanoid_amp.motions.motion_loader", os.path.join(REF, "motions", "motion_loader.py"))
    mpkg.MotionLoader = ml.MotionLoader
    _load("humanoid_amp.g1_cfg", os.path.join(REF, "g1_cfg.py"))
    g1cfg = _load("humanoid_amp.g1_amp_env_cfg", os.path.join(REF, "g1_amp_env_cfg.py"))
    g1env = _load("humanoid_amp.g1_amp_env", os.path.join(REF, "g1_amp_env.py"))
    hcfg = _load("humanoid_amp.humanoid_amp_env_cfg", os.path.join(REF, "humanoid_amp_env_cfg.py"))
    henv = _load("humanoid_amp.humanoid_amp_env", os.path.join(REF, "humanoid_amp_env.py"))
    return ml, g1cfg, g1env, hcfg, henv


def clip(name: str) -> str:
    return os.path.join(REF, "motions", name + ".npz")


# ----------------------------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------------------------


def designed_times(loader, rng: np.random.Generator, n_random: int, k_hist: int):
    """(ids, times): random in-clip times plus every edge case of SURVEY Appendix A.1."""
    n_clips = loader.num_trajectories
    dt = float(loader.dt)
    ids, ts = [], []
    for c in range(n_clips):
        dur = float(loader.durations[c])
        n_c = int(loader.traj_ends[c] - loader.traj_starts[c] + 1)
        special = [0.0, dur, dur - dt, dur - 0.5 * dt, 0.5 * dt, 1.5 * dt, 2.5 * dt, 3.5 * dt, 1e-9, dur * (1 - 1e-12)]
        special += [k * dt for k in (1, 2, 3, 7, n_c // 2, n_c - 2, n_c - 1)]
        special += [(k + 0.5) * dt for k in (4, 5, n_c // 2, n_c - 3)]
        special += [-k * dt for k in range(1, k_hist)]  # history of a random-start reset (blend = -k)
        special += [-0.3 * dt, -2.75 * dt, dur + 0.25 * dt, dur + dt]
        special += list(rng.uniform(0.0, 1.0, size=n_random) * dur)
        ts += special
        ids += [c] * len(special)
    return np.asarray(ids, dtype=np.int64), np.asarray(ts, dtype=np.float64)


def synth_state(rng: np.random.Generator, n: int, n_dof: int, n_bodies: int, ref: int, keys):
    """Synthetic robot.data tensors (SURVEY §8d distributions), AoS like Isaac Lab's views."""
    f32 = np.float32
    body_pos = rng.normal(0.0, 1.0, size=(n, n_bodies, 3)).astype(f32)
    body_pos[:, ref, 2] = rng.uniform(0.35, 0.95, size=n).astype(f32)
    for k in keys:
        body_pos[:, k] = body_pos[:, ref] + rng.normal(0.0, 0.4, size=(n, 3)).astype(f32)
    quat = rng.normal(0.0, 1.0, size=(n, n_bodies, 4))
    quat /= np.linalg.norm(quat, axis=-1, keepdims=True)
    return dict(
        joint_pos=rng.uniform(-1.0, 1.3, size=(n, n_dof)).astype(f32),
        joint_vel=rng.normal(0.0, 1.5, size=(n, n_dof)).astype(f32),
        joint_acc=rng.normal(0.0, 30.0, size=(n, n_dof)).astype(f32),
        body_pos_w=body_pos,
        body_quat_w=quat.astype(f32),
        body_lin_vel_w=rng.normal(0.0, 1.0, size=(n, n_bodies, 3)).astype(f32),
        body_ang_vel_w=rng.normal(0.0, 1.0, size=(n, n_bodies, 3)).astype(f32),
    )


def to_data(state: dict, limits: np.ndarray) -> SimpleNamespace:
    d = SimpleNamespace(**{k: torch.from_numpy(v.copy()) for k, v in state.items()})
    d.soft_joint_pos_limits = torch.from_numpy(limits.copy())
    return d


# ----------------------------------------------------------------------------------------------
# fixture writers
# ----------------------------------------------------------------------------------------------


def gen_motion_fixtures(MotionLoader, tag: str, motion_file: str, k_hist: int, seed: int):
    rng = np.random.default_rng(seed)
    loader = MotionLoader(motion_file, "cpu")
    ids, times = designed_times(loader, rng, n_random=24, k_hist=k_hist)
    i0, i1, blend = loader._compute_frame_blend(times, ids)
    np.savez_compressed(
        os.path.join(OUT, f"frame_blend_{tag}.npz"),
        motion_ids=ids, times=times, index_0=i0.astype(np.int64), index_1=i1.astype(np.int64), blend=blend,
        dt=np.float64(loader.dt), durations=loader.durations, traj_starts=loader.traj_starts, traj_ends=loader.traj_ends,
    )
    # sample: every designed time (50-58 rows per clip)
    pick = np.arange(len(ids))
    s_ids, s_t = ids[pick], times[pick]
    outs = loader.sample(len(pick), times=s_t, motion_ids=s_ids)
    names = ["dof_positions", "dof_velocities", "body_positions", "body_rotations",
             "body_linear_velocities", "body_angular_velocities"]
    np.savez_compressed(
        os.path.join(OUT, f"sample_{tag}.npz"), motion_ids=s_ids, times=s_t,
        **{n: o.numpy() for n, o in zip(names, outs)},
    )
    # default-ids path (times given, motion_ids None -> clip 0) motion_loader.py:365-366
    t0 = s_t[s_ids == 0][:8]
    outs0 = loader.sample(len(t0), times=t0)
    np.savez_compressed(os.path.join(OUT, f"sample_defaultids_{tag}.npz"), times=t0,
                        **{n: o.numpy() for n, o in zip(names, outs0)})
    return loader


def fake_env(loader, cfg, n_envs: int, joint_names, body_names, key_names, n_amp: int, amp_dim: int):
    fake = SimpleNamespace()
    fake.cfg = cfg
    fake.cfg.num_amp_observations = n_amp
    fake.cfg.amp_observation_space = amp_dim
    fake._motion_loader = loader
    fake.num_envs = n_envs
    fake.device = "cpu"
    fake.ref_body_index = body_names.index(cfg.reference_body)
    fake.key_body_indexes = [body_names.index(n) for n in key_names]
    fake.motion_dof_indexes = loader.get_dof_index(joint_names)
    fake.motion_ref_body_index = loader.get_body_index([cfg.reference_body])[0]
    fake.motion_key_body_indexes = loader.get_body_index(key_names)
    fake.amp_observation_size = n_amp * amp_dim
    fake.amp_observation_buffer = torch.zeros((n_envs, n_amp, amp_dim))
    fake.key_body_obs_size = 12
    return fake


def gen_collect_fixture(env_cls, tag: str, fake, seed: int, with_ids: bool):
    rng = np.random.default_rng(seed)
    loader = fake._motion_loader
    n = 32
    ids = rng.integers(0, loader.num_trajectories, size=n).astype(np.int64)
    times = rng.uniform(0.0, 1.0, size=n) * loader.durations[ids]
    times[:4] = [0.0, float(loader.dt) * 0.5, float(loader.dt) * 3, loader.durations[ids[3]]]  # history reaches t<0
    if with_ids:
        out = env_cls.collect_reference_motions(fake, n, times, ids)
    else:
        ids[:] = 0
        times = np.minimum(times, loader.durations[0])
        out = env_cls.collect_reference_motions(fake, n, times)
    np.savez_compressed(
        os.path.join(OUT, f"collect_{tag}.npz"), motion_ids=ids, times=times, amp_obs=out.numpy(),
        num_amp_observations=np.int64(fake.cfg.num_amp_observations),
        motion_dof_indexes=np.asarray(fake.motion_dof_indexes, dtype=np.int64),
        motion_ref_body_index=np.int64(fake.motion_ref_body_index),
        motion_key_body_indexes=np.asarray(fake.motion_key_body_indexes, dtype=np.int64),
    )


def gen_g1_envstep(g1env, tag: str, fake, joint_names, body_names, n_steps: int, seed: int):
    """dones -> rewards -> nonzero -> reset (reference-state init) -> observations, n_steps times."""
    rng = np.random.default_rng(seed)
    np.random.seed(seed)  # MotionLoader.sample_times uses the global legacy RNG (motion_loader.py:321-327)
    torch.manual_seed(seed)
    cfg = fake.cfg
    fake.collect_reference_motions = types.MethodType(g1env.G1AmpEnv.collect_reference_motions, fake)
    N, dof, nb = fake.num_envs, len(joint_names), len(body_names)
    step_dt = (1.0 / 60.0) * cfg.decimation
    fake.max_episode_length = math.ceil(cfg.episode_length_s / step_dt)
    fake.step_dt = step_dt
    lim = np.stack([np.full((N, dof), -0.9 * np.pi / 2), np.full((N, dof), 0.9 * np.pi / 2)], axis=-1).astype(np.float32)
    lim += rng.normal(0, 0.05, size=(1, dof, 2)).astype(np.float32)
    fake.command_target_speed = torch.from_numpy(rng.uniform(-1, 1, size=(N, 2)).astype(np.float32))
    fake.command_time_left = torch.from_numpy(rng.uniform(0.0, 5.0, size=N).astype(np.float32))
    fake.motion_ids = torch.zeros(N, dtype=torch.long)
    fake.motion_start_times = torch.zeros(N)
    fake.last_actions = torch.from_numpy(rng.normal(0, 0.5, size=(N, dof)).astype(np.float32))
    fake.amp_observation_buffer = torch.from_numpy(
        rng.normal(0, 1, size=tuple(fake.amp_observation_buffer.shape)).astype(np.float32))
    fake.scene = SimpleNamespace(env_origins=torch.from_numpy(rng.normal(0, 4.0, size=(N, 3)).astype(np.float32)))
    n_act = cfg.num_actor_observations
    if n_act > 1:
        per = (cfg.amp_observation_space - 12)
        if getattr(cfg, "history_include_last_actions", True):
            per += cfg.action_space
        if getattr(cfg, "history_include_command", True):
            per += 2 if cfg.rew_track_vel > 0.0 else 0
        fake.actor_obs_hist_per_frame = per
        fake.actor_obs_history_buffer = torch.from_numpy(rng.normal(0, 1, size=(N, n_act - 1, per)).astype(np.float32))
        fake._just_reset_mask = torch.zeros(N, dtype=torch.bool)
    rec = {}
    rec["init_amp_observation_buffer"] = fake.amp_observation_buffer.numpy().copy()
    if n_act > 1:
        rec["init_actor_obs_history_buffer"] = fake.actor_obs_history_buffer.numpy().copy()
    rec["soft_joint_pos_limits"] = lim[:1].copy()  # identical for every env (broadcast in the test)
    rec["env_origins"] = fake.scene.env_origins.numpy().copy()
    rec["max_episode_length"] = np.int64(fake.max_episode_length)
    rec["motion_dof_indexes"] = np.asarray(fake.motion_dof_indexes, dtype=np.int64)
    rec["ref_body_index"] = np.int64(fake.ref_body_index)
    rec["key_body_indexes"] = np.asarray(fake.key_body_indexes, dtype=np.int64)
    for s in range(n_steps):
        st = synth_state(rng, N, dof, nb, fake.ref_body_index, fake.key_body_indexes)
        fake.robot = SimpleNamespace(data=to_data(st, lim))
        fake.robot.data.default_root_state = torch.zeros(N, 13)
        fake.actions = torch.from_numpy(rng.normal(0, 0.5, size=(N, dof)).astype(np.float32))
        fake.episode_length_buf = torch.from_numpy(rng.integers(0, fake.max_episode_length + 2, size=N).astype(np.int64))
        p = f"s{s}_"
        for k, v in st.items():
            rec[p + "in_" + k] = v
        rec[p + "in_actions"] = fake.actions.numpy().copy()
        rec[p + "in_episode_length_buf"] = fake.episode_length_buf.numpy().copy()
        rec[p + "in_command_target_speed"] = fake.command_target_speed.numpy().copy()
        # 4. dones (g1_amp_env.py:321-330)
        died, time_out = g1env.G1AmpEnv._get_dones(fake)
        fake.reset_terminated = died
        rec[p + "out_died"] = died.numpy().copy()
        rec[p + "out_time_out"] = time_out.numpy().copy()
        # 5. rewards (g1_amp_env.py:246-319)
        fake.extras = {}
        rew = g1env.G1AmpEnv._get_rewards(fake)
        rec[p + "out_reward"] = rew.numpy().copy()
        for k, v in fake.extras["log"].items():
            rec[p + "log_" + k] = np.float64(v)
        # 6. reset ids: DirectRLEnv.step [recalled]: (terminated | time_outs).nonzero().squeeze(-1)
        ids = (died | time_out).nonzero(as_tuple=False).squeeze(-1)
        rec[p + "out_reset_env_ids"] = ids.numpy().copy()
        if len(ids) > 0:
            start = "start" in cfg.reset_strategy
            rng_state = np.random.get_state()
            m_ids, m_times = fake._motion_loader.sample_times(len(ids), start=start)
            np.random.set_state(rng_state)  # replay the same draw inside the reference call
            root_state, dpos, dvel = g1env.G1AmpEnv._reset_strategy_random(fake, ids, start)
            rec[p + "reset_motion_ids"] = np.asarray(m_ids, dtype=np.int64)
            rec[p + "reset_times"] = np.asarray(m_times, dtype=np.float64)
            rec[p + "out_reset_root_state"] = root_state.numpy().copy()
            rec[p + "out_reset_dof_pos"] = dpos.numpy().copy()
            rec[p + "out_reset_dof_vel"] = dvel.numpy().copy()
            rec[p + "out_reset_amp_rows"] = fake.amp_observation_buffer[ids].numpy().copy()
            # bookkeeping of _reset_idx (g1_amp_env.py:352-358)
            fake.last_actions[ids] = 0.0
            if n_act > 1:
                fake._just_reset_mask[ids] = True
            # emulate write_*_to_sim on the fake sim (our own glue, recorded as INPUT of the obs call)
            d = fake.robot.data
            d.joint_pos[ids] = dpos
            d.joint_vel[ids] = dvel
            d.body_pos_w[ids, fake.ref_body_index] = root_state[:, 0:3]
            d.body_quat_w[ids, fake.ref_body_index] = root_state[:, 3:7]
            d.body_lin_vel_w[ids, fake.ref_body_index] = root_state[:, 7:10]
            d.body_ang_vel_w[ids, fake.ref_body_index] = root_state[:, 10:13]
        # inputs of the obs call that are NOT stored because they are derivable (tests/golden_util.py):
        #   sim state      = in_* with rows[ids] overwritten by out_reset_* (the glue above)
        #   amp buffer     = previous step's out_amp_obs (initial: init_amp_observation_buffer) with
        #                    rows[ids] = out_reset_amp_rows
        #   actor history  = previous step's out_actor_obs_history_buffer (initial: init_...)
        rec[p + "obsin_last_actions"] = fake.last_actions.numpy().copy()
        rec[p + "obsin_command_target_speed"] = fake.command_target_speed.numpy().copy()
        if n_act > 1:
            rec[p + "obsin_just_reset_mask"] = fake._just_reset_mask.numpy().copy()
        # 7. observations (g1_amp_env.py:175-242)
        obs = g1env.G1AmpEnv._get_observations(fake)
        rec[p + "out_policy_obs"] = obs["policy"].numpy().copy()
        rec[p + "out_amp_obs"] = fake.extras["amp_obs"].numpy().copy()
        if n_act > 1:
            rec[p + "out_actor_obs_history_buffer"] = fake.actor_obs_history_buffer.numpy().copy()
        # _apply_action of the next step (g1_amp_env.py:173)
        fake.last_actions = fake.actions.clone()
    rec["n_steps"] = np.int64(n_steps)
    np.savez_compressed(os.path.join(OUT, f"envstep_{tag}.npz"), **rec)


def gen_humanoid_envstep(henv, tag: str, fake, n_dof: int, n_bodies: int, seed: int):
    rng = np.random.default_rng(seed)
    N = fake.num_envs
    fake.max_episode_length = math.ceil(fake.cfg.episode_length_s / ((1.0 / 60.0) * fake.cfg.decimation))
    fake.amp_observation_buffer = torch.from_numpy(
        rng.normal(0, 1, size=tuple(fake.amp_observation_buffer.shape)).astype(np.float32))
    lim = np.zeros((N, n_dof, 2), dtype=np.float32)
    rec = {"init_amp_observation_buffer": fake.amp_observation_buffer.numpy().copy(),
           "max_episode_length": np.int64(fake.max_episode_length),
           "ref_body_index": np.int64(fake.ref_body_index),
           "key_body_indexes": np.asarray(fake.key_body_indexes, dtype=np.int64)}
    for s in range(2):
        st = synth_state(rng, N, n_dof, n_bodies, fake.ref_body_index, fake.key_body_indexes)
        fake.robot = SimpleNamespace(data=to_data(st, lim))
        fake.episode_length_buf = torch.from_numpy(rng.integers(0, fake.max_episode_length + 2, size=N).astype(np.int64))
        p = f"s{s}_"
        for k, v in st.items():
            rec[p + "in_" + k] = v
        rec[p + "in_episode_length_buf"] = fake.episode_length_buf.numpy().copy()
        died, time_out = henv.HumanoidAmpEnv._get_dones(fake)
        rec[p + "out_died"], rec[p + "out_time_out"] = died.numpy().copy(), time_out.numpy().copy()
        rec[p + "out_reset_env_ids"] = (died | time_out).nonzero(as_tuple=False).squeeze(-1).numpy().copy()
        obs = henv.HumanoidAmpEnv._get_observations(fake)
        rec[p + "out_policy_obs"] = obs["policy"].numpy().copy()
        rec[p + "out_amp_obs"] = fake.extras["amp_obs"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"envstep_{tag}.npz"), **rec)


def gen_reward_fn_fixture(g1env, seed: int):
    rng = np.random.default_rng(seed)
    # err^2 on both sides of (and exactly at) floor * sigma^2 = 4 * 0.25 = 1.0
    err = np.concatenate([np.linspace(0, 3, 61), [1.0, np.nextafter(np.float32(1.0), np.float32(2.0)), 0.99999994, 25.0]]).astype(np.float32)
    r = g1env.exp_reward_with_floor(torch.from_numpy(err), 1.0, 0.5, 4.0)
    r2 = g1env.exp_reward_with_floor(torch.from_numpy(err), 0.7, 0.25, 3.0)
    n, dof = 40, 29
    term = torch.from_numpy(rng.integers(0, 2, size=n).astype(bool))
    act = torch.from_numpy(rng.normal(0, 0.5, size=(n, dof)).astype(np.float32))
    jp = torch.from_numpy(rng.uniform(-2.0, 2.0, size=(n, dof)).astype(np.float32))
    lim = np.stack([np.full((n, dof), -1.4), np.full((n, dof), 1.4)], -1).astype(np.float32)
    lim += rng.normal(0, 0.1, size=(1, dof, 2)).astype(np.float32)
    ja = torch.from_numpy(rng.normal(0, 30, size=(n, dof)).astype(np.float32))
    jv = torch.from_numpy(rng.normal(0, 1.5, size=(n, dof)).astype(np.float32))
    total, log = g1env.compute_rewards(-1.0, -0.1, -10.0, -1.0e-06, -0.001, term, act, jp, torch.from_numpy(lim), ja, jv)
    np.savez_compressed(
        os.path.join(OUT, "rewards_fn.npz"), err=err, exp_floor_w1_s05_f4=r.numpy(), exp_floor_w07_s025_f3=r2.numpy(),
        terminated=term.numpy(), actions=act.numpy(), joint_pos=jp.numpy(), soft_joint_pos_limits=lim,
        joint_acc=ja.numpy(), joint_vel=jv.numpy(), total=total.numpy(), **{k: np.float64(v) for k, v in log.items()},
    )


def gen_disc_fixture(tag: str, in_dim: int, seed: int):
    """UNPINNED: skrl is absent.  Restated from SURVEY.md §3.4 / agents/skrl_g1_walk_amp_cfg.yaml:31-39,88-95."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    l1, l2, l3 = torch.nn.Linear(in_dim, 1024), torch.nn.Linear(1024, 512), torch.nn.Linear(512, 1)
    m = 96
    x = torch.randn(m, in_dim, generator=g) * 1.5
    mean = torch.randn(in_dim, generator=g, dtype=torch.float64) * 0.3
    var = torch.rand(in_dim, generator=g, dtype=torch.float64) * 2.0 + 0.05
    task = torch.randn(m, 1, generator=g)
    with torch.no_grad():
        xs = torch.clamp((x - mean.float()) / (torch.sqrt(var.float()) + 1e-8), min=-5.0, max=5.0)
        logits = l3(torch.relu(l2(torch.relu(l1(xs)))))
        style = -torch.log(torch.maximum(1 - 1 / (1 + torch.exp(-logits)), torch.tensor(0.0001)))
        style = style * 2.0
        combined = 0.5 * task + 0.5 * style
        logits64 = (torch.relu(torch.relu(xs.double() @ l1.weight.double().T + l1.bias.double())
                               @ l2.weight.double().T + l2.bias.double()) @ l3.weight.double().T + l3.bias.double())
    np.savez_compressed(
        os.path.join(OUT, f"disc_{tag}.npz"), amp_obs=x.numpy(), running_mean=mean.numpy(), running_variance=var.numpy(),
        seed=np.int64(seed), w1_head=l1.weight.detach().numpy()[0, :8], w3=l3.weight.detach().numpy(), b3=l3.bias.detach().numpy(),
        task_reward=task.numpy(),  # weights: torch.manual_seed(seed); Linear(in,1024), Linear(1024,512), Linear(512,1)
        scaled=xs.numpy(), logits=logits.numpy(), logits_f64=logits64.numpy(), style_reward=style.numpy(), combined=combined.numpy(),
    )


def main():
    ml, g1cfg, g1env, hcfg, henv = import_reference()
    MotionLoader = ml.MotionLoader
    # robot-side joint / body order for G1 = the names stored in G1_dance.npz, identical to the two
    # comment lines at motions/test/get_joint_name.py:231-232 (checked below).
    dance = np.load(clip("G1_dance"))
    g1_joints, g1_bodies = dance["dof_names"].tolist(), dance["body_names"].tolist()
    src = open(os.path.join(REF, "motions", "test", "get_joint_name.py"), encoding="utf-8").read()
    assert str(g1_joints) in src and str(g1_bodies) in src, "robot order mismatch vs get_joint_name.py:231-232"

    # ---- motion fixtures -------------------------------------------------------------------
    walk = gen_motion_fixtures(MotionLoader, "g1_walk", clip("G1_walk"), k_hist=10, seed=101)
    dance_l = gen_motion_fixtures(MotionLoader, "g1_dance", clip("G1_dance"), k_hist=10, seed=102)
    hum3_file = ",".join(clip(n) for n in ("humanoid_walk", "humanoid_run", "humanoid_dance"))
    hum3 = gen_motion_fixtures(MotionLoader, "humanoid3", hum3_file, k_hist=2, seed=103)
    humw = MotionLoader(clip("humanoid_walk"), "cpu")
    hum_joints, hum_bodies = hum3.dof_names, hum3.body_names  # robot order unrecorded -> identity (unpinned)

    # ---- collect_reference_motions ------------------------------------------------------------
    for tag, loader, K in (("g1_walk_k2", walk, 2), ("g1_walk_k10", walk, 10), ("g1_dance_k10", dance_l, 10)):
        fake = fake_env(loader, g1cfg.G1AmpEnvCfg_CUSTOM(), 32, g1_joints, g1_bodies, G1_KEY_BODIES, K, 83)
        gen_collect_fixture(g1env.G1AmpEnv, tag, fake, seed=200 + K, with_ids=True)
    fake = fake_env(hum3, hcfg.HumanoidAmpEnvCfg(), 32, hum_joints, hum_bodies, HUM_KEY_BODIES, 2, 81)
    gen_collect_fixture(g1env.G1AmpEnv, "humanoid3_k2", fake, seed=210, with_ids=True)  # G1 calling convention
    # HumanoidAmpEnv.collect_reference_motions (humanoid_amp_env.py:219-248) cannot be pinned: in this fork it
    # calls sample(num_samples=n, times=<n*K values>) without motion_ids, and MotionLoader.sample then builds
    # np.zeros(n) ids (motion_loader.py:365-366) -> numpy broadcast ValueError for every K > 1 (observed here).
    # The humanoid clips are therefore pinned through the G1 calling convention only (fixture above).
    try:
        fake = fake_env(humw, hcfg.HumanoidAmpEnvCfg(), 32, hum_joints, hum_bodies, HUM_KEY_BODIES, 2, 81)
        henv.HumanoidAmpEnv.collect_reference_motions(fake, 4, np.zeros(4))
        raise SystemExit("reference humanoid collect_reference_motions unexpectedly works: add a fixture for it")
    except ValueError:
        pass

    # ---- env step sequences --------------------------------------------------------------------
    cfg = g1cfg.G1AmpDanceEnvCfg()  # K=10, D=83, rewards on, random-start, single-frame actor obs (102)
    fake = fake_env(dance_l, cfg, 24, g1_joints, g1_bodies, G1_KEY_BODIES, 10, 83)
    gen_g1_envstep(g1env, "g1_dance_custom", fake, g1_joints, g1_bodies, n_steps=3, seed=301)

    cfg = g1cfg.G1AmpEnvCfg_CUSTOM()  # walk clip with the self-consistent custom cfg, K=2, random reset
    cfg.reset_strategy = "random"
    fake = fake_env(walk, cfg, 32, g1_joints, g1_bodies, G1_KEY_BODIES, 2, 83)
    gen_g1_envstep(g1env, "g1_walk_k2", fake, g1_joints, g1_bodies, n_steps=3, seed=302)

    cfg = g1cfg.G1AmpDeployEnvCfg()  # actor-observation history (3 frames) with warm start
    cfg.num_actor_observations = 3
    cfg.__post_init__()
    cfg.rew_termination, cfg.rew_action_l2 = -1.0, -0.1
    fake = fake_env(walk, cfg, 24, g1_joints, g1_bodies, G1_KEY_BODIES, 2, 83)
    gen_g1_envstep(g1env, "g1_deploy_hist3", fake, g1_joints, g1_bodies, n_steps=3, seed=303)
    rec_obs_space = np.int64(cfg.observation_space)

    cfg = g1cfg.G1AmpDeployEnvCfg()  # ablation: history frames carry only the base obs
    cfg.num_actor_observations = 2
    cfg.history_include_last_actions = False
    cfg.history_include_command = False
    cfg.__post_init__()
    fake = fake_env(walk, cfg, 24, g1_joints, g1_bodies, G1_KEY_BODIES, 2, 83)
    gen_g1_envstep(g1env, "g1_deploy_hist2_ablate", fake, g1_joints, g1_bodies, n_steps=3, seed=304)

    fake = fake_env(hum3, hcfg.HumanoidAmpEnvCfg(), 32, hum_joints, hum_bodies, HUM_KEY_BODIES, 2, 81)
    gen_humanoid_envstep(henv, "humanoid3", fake, 28, len(hum_bodies), seed=305)

    gen_reward_fn_fixture(g1env, seed=400)
    gen_disc_fixture("k2_166", 166, seed=0)

    np.savez_compressed(os.path.join(OUT, "meta.npz"), g1_robot_joint_names=np.asarray(g1_joints),
                        g1_robot_body_names=np.asarray(g1_bodies), deploy_hist3_observation_space=rec_obs_space)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
