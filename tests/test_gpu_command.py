"""GPU parity: velocity-command timers (g1_amp_env.py:146-167,421-439) and the lazily read reward log.

The reference resamples with torch.rand on the global CUDA generator behind a nonzero() host sync; the engine draws
with a counter-based Philox stream keyed (seed, step, global env id), so parity with the REFERENCE is distributional by
construction.  What is bit-exact: the engine vs oracle/rng.py (itself pinned by the Random123 known answers), the timer
arithmetic, which envs are resampled, the fixed-command branch, and invariance under sharding.
"""

import numpy as np
import pytest
import torch

from oracle import rng as orng

pytestmark = pytest.mark.gpu

VEL, TIME, DT = (-1.0, 1.0), (4.0, 7.0), 1.0 / 30.0


def _state(N, seed):
    g = np.random.default_rng(seed)
    cmd = g.uniform(-1, 1, (N, 2)).astype(np.float32)
    # a mix of running, about-to-expire, exactly-expiring, expired and infinite timers
    left = g.uniform(-0.05, 0.3, N).astype(np.float32)
    left[::7] = np.float32(DT)
    left[3::11] = np.inf
    return cmd, left


@pytest.mark.parametrize("N", [1, 255, 256, 257, 5000, 65536])
@pytest.mark.parametrize("vel", [VEL, (0.5, 0.5)])
def test_command_tick_bit_exact(N, vel):
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import command_step

    cmd, left = _state(N, N)
    c_d, l_d = torch.from_numpy(cmd).cuda(), torch.from_numpy(left).cuda()
    command_step(c_d, l_d, mode=nat.AMP_COMMAND_TICK, step_dt=DT, vel_range=vel, time_range=TIME, seed=(7 << 32) | 99, step=1234,
                 env_offset=40000)
    c_o, l_o = orng.command_tick(cmd, left, DT, vel, TIME, seed=(7 << 32) | 99, step=1234, env_offset=40000)
    assert np.array_equal(c_d.cpu().numpy(), c_o) and np.array_equal(l_d.cpu().numpy(), l_o)
    if vel[1] > vel[0]:
        resampled = (left - np.float32(DT)) <= 0
        assert resampled.sum() > 0 or N == 1
        assert (l_o[resampled] >= 4.0).all() and (l_o[resampled] < 7.0).all()


@pytest.mark.parametrize("vel", [VEL, (0.5, 0.5)])
def test_command_reset_mask_and_id_list_agree(vel):
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import command_step

    N = 3001
    cmd, left = _state(N, 5)
    mask = np.random.default_rng(1).uniform(size=N) < 0.2
    ids = np.nonzero(mask)[0]
    want_c, want_l = orng.command_reset(cmd, left, ids, vel, TIME, seed=3, step=17, env_offset=8192)
    # (a) mask form (device-reset path)
    c_d, l_d = torch.from_numpy(cmd).cuda(), torch.from_numpy(left).cuda()
    command_step(c_d, l_d, mode=nat.AMP_COMMAND_RESET, step_dt=DT, vel_range=vel, time_range=TIME, seed=3, step=17, env_offset=8192,
                 reset_mask=torch.from_numpy(mask).cuda())
    assert np.array_equal(c_d.cpu().numpy(), want_c) and np.array_equal(l_d.cpu().numpy(), want_l)
    # (b) id list capped by a device-side count (ids beyond the count are garbage on purpose)
    ids_d = torch.full((N,), N + 5, dtype=torch.int64).cuda()
    ids_d[: len(ids)] = torch.from_numpy(ids).cuda()
    c_d, l_d = torch.from_numpy(cmd).cuda(), torch.from_numpy(left).cuda()
    command_step(c_d, l_d, mode=nat.AMP_COMMAND_RESET, step_dt=DT, vel_range=vel, time_range=TIME, seed=3, step=17, env_offset=8192,
                 env_ids=ids_d, count=torch.tensor([len(ids)], dtype=torch.int64).cuda())
    assert np.array_equal(c_d.cpu().numpy(), want_c) and np.array_equal(l_d.cpu().numpy(), want_l)


def test_command_draw_is_shard_invariant_and_uniform():
    """Two shards [0, 3000) and [3000, 8000) with their env_offset reproduce the unsharded 8000-env draw; the draw is
    uniform on [lo, hi) x [lo, hi) x [t_lo, t_hi) (the reference's distribution, g1_amp_env_cfg.py:96-100)."""
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import command_step

    N = 200000
    c = torch.zeros(N, 2).cuda()
    l = torch.zeros(N).cuda()  # every timer expires
    command_step(c, l, mode=nat.AMP_COMMAND_TICK, step_dt=DT, vel_range=VEL, time_range=TIME, seed=21, step=2)
    parts_c, parts_l = [], []
    for lo, hi in ((0, 3000), (3000, 8000)):
        pc, pl = torch.zeros(hi - lo, 2).cuda(), torch.zeros(hi - lo).cuda()
        command_step(pc, pl, mode=nat.AMP_COMMAND_TICK, step_dt=DT, vel_range=VEL, time_range=TIME, seed=21, step=2, env_offset=lo)
        parts_c.append(pc), parts_l.append(pl)
    assert torch.equal(torch.cat(parts_c), c[:8000]) and torch.equal(torch.cat(parts_l), l[:8000])
    cn, ln = c.cpu().numpy(), l.cpu().numpy()
    assert -1.0 <= cn.min() and cn.max() < 1.0 and 4.0 <= ln.min() and ln.max() < 7.0
    assert abs(cn.mean()) < 5e-3 and abs(cn.var() - 1 / 3) < 5e-3 and abs(ln.mean() - 5.5) < 1e-2 and abs(ln.var() - 0.75) < 1e-2


def test_reward_log_means_and_lazy_log():
    from humanoid_amp_amd.engine import REWARD_TERMS, LazyRewardLog, reward_log_means

    terms = torch.randn(len(REWARD_TERMS), 65536, generator=torch.Generator().manual_seed(0)) * torch.arange(1, 9)[:, None]
    means = reward_log_means(terms.cuda())
    want = terms.double().mean(dim=1)
    assert float((means.cpu().double() - want).abs().max()) <= 1e-6
    log = LazyRewardLog(REWARD_TERMS, means, drop=("rew_track_vel", "error_track_vel"))
    assert not log.materialized
    assert "rew_track_vel" not in log and log.materialized and len(log) == 6
    assert abs(log["total_reward"] - float(want[0])) <= 1e-6 and isinstance(log["total_reward"], float)
    assert dict(log.items()).keys() == {k for k in REWARD_TERMS if "track" not in k}
    # mutators act on the materialised values, like the plain dict the reference hands to skrl (never on the empty shell)
    fresh = lambda: LazyRewardLog(REWARD_TERMS, means, drop=("rew_track_vel", "error_track_vel"))  # noqa: E731
    l2 = fresh()
    assert abs(l2.pop("total_reward") - float(want[0])) <= 1e-6 and len(l2) == 5
    l3 = fresh()
    l3.update(total_reward=7.0, extra=1.0)
    assert l3["total_reward"] == 7.0 and l3["extra"] == 1.0 and len(l3) == 7
    l4 = fresh()
    assert l4.setdefault("pub_action_l2", -1.0) != -1.0 and (fresh() | {"x": 1})["x"] == 1
    assert fresh() == fresh() and not (fresh() != fresh()) and fresh() == dict(fresh().items())
    assert list(reversed(fresh())) == list(reversed(list(fresh())))
    k, v = fresh().popitem()
    assert k == "pub_joint_vel_l2" and isinstance(v, float)
    # a ragged env count takes the scalar tail / the unaligned-row path
    for n in (4099, 37):
        t2 = torch.randn(3, n, generator=torch.Generator().manual_seed(n))
        got = reward_log_means(t2.cuda())
        assert float((got.cpu().double() - t2.double().mean(dim=1)).abs().max()) <= 1e-6


def test_command_draws_follow_the_run_seed_and_the_shard():
    """the velocity-command stream is keyed by the run's seed (torch.manual_seed / cfg.seed / env.reset(seed=)),
    as the reference's torch.rand draws are, and two ranks never draw the same stream for their local env i."""
    from humanoid_amp_amd.envs import G1AmpEnv, G1AmpEnvCfg_CUSTOM, G1AmpWalkEnvCfg

    def run(manual_seed=None, reset_seed=None, env_offset=None, ctor_seed=None, cfg_seed=None):
        cfg = G1AmpEnvCfg_CUSTOM(motion_file=G1AmpWalkEnvCfg().motion_file, num_amp_observations=2, reset_strategy="random")
        cfg.scene.num_envs = 256
        if cfg_seed is not None:
            cfg.seed = cfg_seed
        if manual_seed is not None:
            torch.manual_seed(manual_seed)
        env = G1AmpEnv(cfg, device_reset=True, reset_seed=ctor_seed, env_offset=env_offset)
        env.reset(seed=reset_seed)
        for _ in range(3):
            env.step(torch.zeros(256, 29, device="cuda"))
        return env.command_target_speed.clone()

    assert torch.equal(run(manual_seed=1), run(manual_seed=1))
    assert not torch.equal(run(manual_seed=1), run(manual_seed=2))          # torch.manual_seed before construction
    assert not torch.equal(run(manual_seed=1, reset_seed=10), run(manual_seed=1, reset_seed=11))   # env.reset(seed=)
    assert torch.equal(run(manual_seed=1, reset_seed=10), run(manual_seed=2, reset_seed=10))
    assert not torch.equal(run(cfg_seed=4), run(cfg_seed=5)) and torch.equal(run(cfg_seed=4), run(manual_seed=9, cfg_seed=4))
    assert not torch.equal(run(ctor_seed=3, env_offset=0), run(ctor_seed=3, env_offset=256))       # rank 0 vs rank 1 shards


def test_g1_env_step_has_no_host_sync():
    """device_reset=True: a whole G1 step -- command timers with a non-empty range, dones, rewards WITH logging, device
    reset, observations -- under torch.cuda.set_sync_debug_mode("error"): any .item() / nonzero / blocking copy inside
    step() raises.  The log is read afterwards, outside the guarded region."""
    from humanoid_amp_amd.envs import G1AmpEnv, G1AmpEnvCfg_CUSTOM, G1AmpWalkEnvCfg

    cfg = G1AmpEnvCfg_CUSTOM(motion_file=G1AmpWalkEnvCfg().motion_file, num_amp_observations=2, reset_strategy="random")
    assert cfg.track_vel_range[1] > cfg.track_vel_range[0] and cfg.rew_track_vel > 0.0
    cfg.scene.num_envs = 512
    cfg.episode_length_s = 0.2
    env = G1AmpEnv(cfg, device_reset=True, reset_seed=3, log_rewards=True)
    env.reset()
    acts = [torch.randn(512, 29, device="cuda") * 0.3 for _ in range(12)]
    for a in acts[:2]:
        env.step(a)  # warm-up: lazy allocations
    torch.cuda.synchronize()
    resets, logs = torch.zeros((), dtype=torch.int64, device="cuda"), []
    cmd_before = env.command_target_speed.clone()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for a in acts[2:]:
            obs, rew, term, tout, extras = env.step(a)
            resets += (term | tout).sum()
            logs.append(extras["log"])
            assert not extras["log"].materialized
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert int(resets) >= 512                      # time-outs happened: the device reset path ran inside the guard
    assert not torch.equal(cmd_before, env.command_target_speed)  # commands were resampled on the device
    c = env.command_target_speed
    assert float(c.min()) >= cfg.track_vel_range[0] and float(c.max()) < cfg.track_vel_range[1]
    assert float(env.command_time_left.max()) < cfg.command_resampling_time_range[1]
    assert set(logs[-1]) == {"total_reward", "rew_track_vel", "error_track_vel", "pub_termination", "pub_action_l2",
                             "pub_joint_pos_limits", "pub_joint_acc_l2", "pub_joint_vel_l2"}
    assert all(np.isfinite(v) for v in logs[-1].values())
    # the per-env mirrors of the reset draw (g1_amp_env.py:377-382) follow on the device path too
    assert int(env.motion_ids.max()) == 0 and float(env.motion_start_times.max()) > 0.0


def test_pre_physics_step_equals_the_separate_launches():
    """amp_pre_physics_step (actions copy + joint targets + last_actions + command tick in ONE launch) against the five
    ATen launches of the reference's _pre_physics_step / _apply_action (g1_amp_env.py:142-173) + amp_command_step(TICK):
    torch.equal on everything, ragged env count, NULL outputs skipped."""
    import ctypes as C

    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import command_step

    N, A = 5003, 29
    gen = torch.Generator().manual_seed(8)
    acts = (torch.randn(N, A, generator=gen) * 0.7).cuda()
    off, scale = torch.randn(A, generator=gen).cuda(), (torch.rand(A, generator=gen) + 0.5).cuda()
    cmd0, left0 = (torch.rand(N, 2, generator=gen) * 2 - 1).cuda(), (torch.rand(N, generator=gen) * 0.1 - 0.02).cuda()
    # separate launches
    cmd_a, left_a = cmd0.clone(), left0.clone()
    command_step(cmd_a, left_a, mode=nat.AMP_COMMAND_TICK, step_dt=DT, vel_range=VEL, time_range=TIME, seed=77, step=5, env_offset=1000)
    want_target = off + scale * acts
    # one launch
    cmd_b, left_b = cmd0.clone(), left0.clone()
    actions, last, target = torch.zeros_like(acts), torch.zeros_like(acts), torch.zeros_like(acts)
    a = nat.AmpPrePhysicsArgs()
    a.actions_in, a.actions, a.last_actions, a.target = acts.data_ptr(), actions.data_ptr(), last.data_ptr(), target.data_ptr()
    a.offset, a.scale, a.num_envs, a.n_actions = off.data_ptr(), scale.data_ptr(), N, A
    ep = torch.randint(0, 300, (N,), generator=gen).cuda()   # DirectRLEnv.step's episode_length_buf += 1 rides on the launch
    ep0 = ep.clone()
    a.episode_length = ep.data_ptr()
    t = nat.AmpCommandArgs()
    t.command, t.time_left = cmd_b.data_ptr(), left_b.data_ptr()
    t.step_dt, t.vel_lo, t.vel_span, t.t_lo, t.t_span = DT, VEL[0], VEL[1] - VEL[0], TIME[0], TIME[1] - TIME[0]
    t.seed, t.step, t.env_offset = 77, 5, 1000
    with torch.cuda.device("cuda:0"):
        nat.check(nat.load().amp_pre_physics_step(C.byref(a), C.byref(t), nat.stream_ptr()), "amp_pre_physics_step")
    assert torch.equal(actions, acts) and torch.equal(last, acts) and torch.equal(target, want_target)
    assert torch.equal(ep, ep0 + 1)
    assert torch.equal(cmd_a, cmd_b) and torch.equal(left_a, left_b) and not torch.equal(cmd0, cmd_b)
    # NULL outputs / no tick / no affine map
    a2 = nat.AmpPrePhysicsArgs()
    a2.actions_in, a2.target, a2.num_envs, a2.n_actions = acts.data_ptr(), target.data_ptr(), N, A
    with torch.cuda.device("cuda:0"):
        nat.check(nat.load().amp_pre_physics_step(C.byref(a2), None, nat.stream_ptr()), "amp_pre_physics_step")
    assert torch.equal(target, acts) and torch.equal(cmd_a, cmd_b)
    # arrays that are not 16-B aligned take the one-element-per-lane body: same results
    n3 = 301
    pool = torch.zeros(3 * n3 * A + 8, device="cuda")
    src = pool[1 : 1 + n3 * A].view(n3, A)
    src.copy_(acts[:n3])
    dst_t, dst_l = pool[n3 * A + 3 : 2 * n3 * A + 3].view(n3, A), pool[2 * n3 * A + 6 : 3 * n3 * A + 6].view(n3, A)
    a3 = nat.AmpPrePhysicsArgs()
    a3.actions_in, a3.target, a3.last_actions = src.data_ptr(), dst_t.data_ptr(), dst_l.data_ptr()
    a3.offset, a3.scale, a3.num_envs, a3.n_actions = off.data_ptr(), scale.data_ptr(), n3, A
    with torch.cuda.device("cuda:0"):
        nat.check(nat.load().amp_pre_physics_step(C.byref(a3), None, nat.stream_ptr()), "amp_pre_physics_step")
    assert torch.equal(dst_t, want_target[:n3]) and torch.equal(dst_l, acts[:n3])
    # fewer actions than a quad (the joint index wraps inside one lane's four elements)
    a_s = (torch.randn(64, 3, generator=gen)).cuda()
    o_s, s_s = torch.randn(3, generator=gen).cuda(), torch.randn(3, generator=gen).cuda()
    t_s = torch.zeros_like(a_s)
    a4 = nat.AmpPrePhysicsArgs()
    a4.actions_in, a4.target, a4.offset, a4.scale, a4.num_envs, a4.n_actions = a_s.data_ptr(), t_s.data_ptr(), o_s.data_ptr(), s_s.data_ptr(), 64, 3
    with torch.cuda.device("cuda:0"):
        nat.check(nat.load().amp_pre_physics_step(C.byref(a4), None, nat.stream_ptr()), "amp_pre_physics_step")
    assert torch.equal(t_s, o_s + s_s * a_s)
