"""GPU: HIP-path shard equivalence (last bullet; 8e): env-sharded runs reproduce the single-GPU run bit for bit.

BASELINE.json configs[3..4] shard 32 768 / 65 536 envs over 4 / 8 GPUs (8 192 envs per GPU; reference caller
train.py:183-196).  An 8 192-env shard takes different code than the 65 536-env whole: 16-env tiles instead of 32 in
amp_env_step, the small-shard GEMM kernels instead of the 256 x 256 LDS-DMA ones, one row chunk instead of two.  On
ONE GPU, run the whole and then every [lo, hi) block of `shard_bounds` for world 2 / 4 / 8 (uneven worlds too) on the
matching rows of the same synthetic state, and require the concatenation to equal the whole: AMP history, policy obs,
expert rows, task reward, done bits, global reset ids, style / combined rewards -- all `torch.equal`.  The collective
curve itself (N > 1 ranks over RCCL) is measured by the driver only; this test pins that sharding cannot change results.
"""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(spec, n, state, steps=2):
    import contextlib, io

    from humanoid_amp_amd.workloads import HotPath

    with contextlib.redirect_stdout(io.StringIO()):
        hot = HotPath(spec, n, "cuda:0", state=state, log_reward_terms=True)
    outs = []
    for _ in range(steps):
        o = hot.step()
        k = hot.kernel
        cnt = int(k.reset_count.item())
        outs.append(dict(amp=k.amp_observation_buffer.clone(), pol=k.policy_obs.clone(), rew=k.reward.clone(), died=k.died.clone(),
                         tout=k.time_out.clone(), ids=k.reset_ids[:cnt].clone(), expert=hot.expert_obs.clone(),
                         style=o["style"].clone(), comb=o["combined"].clone(), terms=k.reward_terms.clone()))
    return outs


@pytest.mark.parametrize("workload,total,worlds", [("g1_walk", 65536, (8, 4, 2)), ("humanoid3", 32768, (4,)),
                                                    ("g1_walk", 20000, (3, 7)), ("g1_dance", 8192, (2,))])
def test_shards_concatenate_to_the_whole(workload, total, worlds):
    from humanoid_amp_amd.distributed import global_env_ids, shard_bounds
    from humanoid_amp_amd.motions import MOTIONS_DIR, MotionLoader
    from humanoid_amp_amd.synthetic import make_state
    from humanoid_amp_amd.workloads import WORKLOADS
    import contextlib, io, os

    spec = WORKLOADS[workload]
    with contextlib.redirect_stdout(io.StringIO()):
        dur = MotionLoader(",".join(os.path.join(MOTIONS_DIR, c + ".npz") for c in spec.clips), "cuda:0").durations
    state = make_state(total, spec.n_dof, spec.max_episode_length, dur, 4242, "cuda:0")
    whole = _run(spec, total, state)
    assert all(len(w["ids"]) > 0 for w in whole)
    for world in worlds:
        parts = []
        for rank in range(world):
            lo, hi = shard_bounds(total, world, rank)
            sl = {k: (v if k == "soft_limits" else v[lo:hi].contiguous()) for k, v in state.items()}
            part = _run(spec, hi - lo, sl)
            for p in part:
                p["ids"] = global_env_ids(p["ids"], total, world, rank)
            parts.append(part)
        for step, w in enumerate(whole):
            for key in w:
                dim = 1 if key == "terms" else 0
                got = torch.cat([parts[r][step][key] for r in range(world)], dim=dim)
                assert got.shape == w[key].shape, (world, step, key, got.shape, w[key].shape)
                assert torch.equal(got, w[key]), (workload, world, step, key, float((got.float() - w[key].float()).abs().max()))


def test_device_reset_and_commands_are_shard_invariant():
    """The counter-based draws (reset clip / time, velocity commands) are keyed by the GLOBAL env id: a shard with its
    env_offset draws what the unsharded run draws for the same envs (g1_amp_env.py:371-439 runs per env)."""
    import contextlib, io, os

    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.distributed import shard_bounds
    from humanoid_amp_amd.engine import command_step
    from humanoid_amp_amd.motions import MOTIONS_DIR, MotionLoader
    from humanoid_amp_amd.robots import G1_JOINT_NAMES, G1_KEY_BODY_NAMES

    N, K = 4096, 2
    with contextlib.redirect_stdout(io.StringIO()):
        ml = MotionLoader(os.path.join(MOTIONS_DIR, "G1_walk.npz"), "cuda:0")
    ml.set_obs_layout(ml.get_dof_index(G1_JOINT_NAMES), ml.get_body_index(["pelvis"])[0], ml.get_body_index(G1_KEY_BODY_NAMES))
    D = ml.obs_size
    mask = torch.rand(N, generator=torch.Generator().manual_seed(3)) < 0.3
    origins = torch.randn(N, 3, generator=torch.Generator().manual_seed(4)).cuda()

    def reset(lo, hi):
        n = hi - lo
        ids = mask[lo:hi].nonzero().squeeze(-1).cuda()
        pad = torch.zeros(n, dtype=torch.int64, device="cuda")
        pad[: len(ids)] = ids
        buf = torch.zeros(n, K, D, device="cuda")
        m_ids = torch.full((n,), -1, dtype=torch.int64, device="cuda")
        m_t = torch.full((n,), -1.0, device="cuda")
        out = ml.reset_apply(pad, torch.tensor([len(ids)], device="cuda"), K, seed=9, step=5, start=False,
                             env_origins=origins[lo:hi].contiguous(), z_lift=0.05, amp_observation_buffer=buf,
                             env_motion_ids=m_ids, env_motion_start_times=m_t, env_offset=lo)
        cmd, left = torch.zeros(n, 2, device="cuda"), torch.zeros(n, device="cuda")
        command_step(cmd, left, mode=nat.AMP_COMMAND_RESET, step_dt=1 / 30, vel_range=(-1.0, 1.0), time_range=(4.0, 7.0), seed=9,
                     step=5, env_offset=lo, reset_mask=mask[lo:hi].cuda())
        c = len(ids)
        return dict(buf=buf, m_ids=m_ids, m_t=m_t, cmd=cmd, left=left, root=out["root_state"][:c].clone(), dof=out["dof_pos"][:c].clone())

    whole = reset(0, N)
    assert int((whole["m_ids"] >= 0).sum()) == int(mask.sum()) and float(whole["m_t"].max()) > 0.0
    for world in (2, 8):
        parts = [reset(*shard_bounds(N, world, r)) for r in range(world)]
        for key in whole:
            assert torch.equal(torch.cat([p[key] for p in parts]), whole[key]), (world, key)


@pytest.mark.parametrize("envs", [1000, 8192, 12288, 20000, 40000])
def test_fused_tail_and_every_kernel_plan_agree(envs):
    """(a) amp_disc_style_reward_prescaled_compact (compaction riding on the finalize launch) == the two separate
    launches, bit for bit; (b) the kernel plans a shard size selects (register-staged 64 x 64, LDS-DMA 128 x 128,
    256 x 256 + 256 x 128, 256 x 256 in one or two chunks) give the same style rewards as the fp32-MFMA engine to
    1e-6 and -- through test_shards_concatenate_to_the_whole -- the same bits as each other."""
    import contextlib, io

    from humanoid_amp_amd.workloads import WORKLOADS, HotPath

    spec = WORKLOADS["g1_walk"]
    outs = {}
    for name, kw in (("fused", dict(fused_tail=True)), ("separate", dict(fused_tail=False)), ("f32", dict(disc_precision="f32"))):
        with contextlib.redirect_stdout(io.StringIO()):
            hot = HotPath(spec, envs, "cuda:0", seed=5, **kw)
        o = hot.step()
        o = hot.step()
        n = int(hot.kernel.reset_count.item())
        outs[name] = (o["style"].clone(), o["combined"].clone(), hot.kernel.reset_ids[:n].clone())
    for a, b in zip(outs["fused"], outs["separate"]):
        assert torch.equal(a, b)
    assert len(outs["fused"][2]) > 0 and torch.equal(outs["fused"][2], outs["f32"][2])
    assert float((outs["fused"][0] - outs["f32"][0]).abs().max()) <= 2e-6
