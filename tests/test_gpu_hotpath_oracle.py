"""GPU: the TIMED path meets the oracle in one hop.

``HotPath.step()`` (= ``amp_hot_step``: LDS-DMA env body with rewards, 256-sample / 64-sample expert body, fused scaler,
layer 1 / layer 2, fused tail with the reset-id compaction) is run for three steps on three different synthetic states and
every output of every step is compared with ``oracle/hotpath.py`` -- the composition of oracle/{motion,env,disc}.py --
which keeps its OWN AMP history across the steps, so drift would show.  Sizes cover the tile plans the benchmark uses:
32-env tiles are exercised at 20 000 envs (>= 16 384), 16-env tiles at 4 096 / 3 000 (ragged last tile), 8-env tiles with
K = 10 at 1 000; the headline's discriminator plan since round 4 -- the fused two-layer kernel -- at 36 000 envs (one full round
of 128-row tiles + a 3 232-row remainder on the two-kernel plan) and at 25 000 humanoid envs (one ragged round).  Bars as in ``__graft_entry__.smoke()``: done bits and reset ids bit-exact, everything else <= 1e-5
(task / combined reward relative to max(1, |reward|); discriminator: skrl absent, parity unpinned).
"""

import os

import pytest
import torch

from oracle import hotpath as ohot
from oracle import motion as om

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("workload,num_envs", [("g1_walk", 4096), ("g1_walk", 20000), ("g1_dance", 1000), ("humanoid3", 3000),
                                               ("g1_walk", 36000), ("humanoid3", 25000), ("g1_walk_23dof", 4096), ("g1_walk_23dof", 30000)])
def test_hot_step_matches_the_oracle(workload, num_envs):
    from humanoid_amp_amd.robots import G1_JOINT_NAMES, G1_KEY_BODY_NAMES, HUMANOID_KEY_BODY_NAMES
    from humanoid_amp_amd.workloads import WORKLOADS, HotPath, clip_files

    spec = WORKLOADS[workload]
    hot = HotPath(spec, num_envs, "cuda:0", seed=3, state_sets=3)
    assert hot.one_call and (hot.fused_scaler or hot.raw_rows) and hot.fused_tail and hot.fused_expert  # the benchmark's configuration
    # (g1_walk_23dof: BASELINE configs[1] read literally -- a synthetic 23-DoF variant of the clip, D = 71, K D = 142: the 160-wide
    #  two-kernel plan at 4 096 envs, 5 k-blocks of activation fragments in the fused two-layer kernel at 30 000)
    mt = om.load_tables(clip_files(spec))
    lay = ohot.layout(mt, spec.robot, spec.joint_names if spec.robot == "g1" else G1_JOINT_NAMES, G1_KEY_BODY_NAMES, HUMANOID_KEY_BODY_NAMES)
    shadow = hot.kernel.amp_observation_buffer.cpu().clone()  # the oracle's own history from here on
    tile = hot.kernel.tile_envs
    assert tile == (8 if spec.K == 10 else (32 if num_envs >= 16384 else 16))
    total_resets = 0
    for step in range(3):
        st = {k: v.cpu() for k, v in hot.states[step].items()}
        out = hot.step()
        torch.cuda.synchronize()
        exp = ohot.step(mt, lay, spec, dict(hot.cfg.__dict__), st, shadow, hot.disc_weights, max_episode_length=spec.max_episode_length)
        if step == 0:  # the initial history was the expert rows of state 0
            assert float((exp["expert"] - hot.expert_obs.cpu()).abs().max()) <= TOL
        err = ohot.compare(hot, out, exp)
        assert err["dones_equal"] and err["reset_ids_equal"], (step, err)
        scale = max(1.0, float(exp["task"].abs().max()))
        assert err["expert"] <= TOL and err["amp"] <= TOL and err["policy"] <= TOL, (step, err)
        assert err["task"] <= TOL * scale and err["style"] <= TOL and err["combined"] <= TOL * scale, (step, err)
        total_resets += err["n_reset"]
    assert total_resets > 0


def test_hot_args_follow_a_reallocated_workspace():
    """amp_hot_step runs on prebuilt arguments holding raw device pointers.  A later call on the same discriminator
    that needs a LARGER workspace replaces the tensor: the next hot step must rebuild its arguments instead of writing through
    the stale address.  Two identical shards; one of them is disturbed between the steps; results stay bit-equal."""
    from humanoid_amp_amd.workloads import WORKLOADS, HotPath

    spec = WORKLOADS["g1_walk"]
    a = HotPath(spec, 512, "cuda:0", seed=5, state_sets=2)
    b = HotPath(spec, 512, "cuda:0", seed=5, state_sets=2)
    oa, ob = a.step(), b.step()
    assert torch.equal(oa["style"], ob["style"])
    ws_before = b.disc._ws[0].data_ptr()
    big = torch.randn(8192, spec.K * spec.D, device="cuda")
    b.disc.style_reward(big)                       # 16x the rows: the workspace of slot 0 is reallocated
    assert b.disc._ws[0].data_ptr() != ws_before
    junk = torch.full((1 << 22,), float("nan"), device="cuda")  # whatever reuses the freed block must not be read or written
    oa, ob = a.step(), b.step()
    torch.cuda.synchronize()
    assert torch.equal(oa["style"], ob["style"]) and torch.equal(oa["combined"], ob["combined"])
    assert torch.equal(a.kernel.amp_observation_buffer, b.kernel.amp_observation_buffer)
    assert bool(torch.isnan(junk).all())
