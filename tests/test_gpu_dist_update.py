"""GPU: the multi-rank discriminator update (engine.AmpDiscriminatorUpdate(group=...) over distributed.UpdateExchange).

The reference's --distributed mode keeps one agent replica per GPU in step (train.py:54-58,183-196; skrl's gradient all-reduce is
third-party: parity unpinned).  Here the replicas stay in step because every rank takes the same optimizer steps on the same
all-gathered global minibatches.  Two ranks share the ONE visible GPU (gloo rendezvous; the device tensors travel through the
host on that rehearsal path -- the RCCL path itself is tests/test_gpu_dist.py, world 1) and must end bit-identical, and equal to
a single process stepped on the concatenated rows."""

import os
import socket
import subprocess
import sys

import pytest
import torch

from oracle import disc as odisc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_gpu_stay_bit_identical_and_equal_a_single_process(tmp_path):
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_update_worker as wk

    world = 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_update_worker.py"), str(tmp_path)],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    a, b = (torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(world))

    n_steps = wk.UPDATES * wk.EPOCHS * wk.MBS
    assert a["step"] == b["step"] == n_steps and a["rows_per_rank"] == wk.BS // world
    # (1) the replicas: weights, Adam moments, scaler statistics, losses, refreshed inference planes -- bit-identical
    assert all(torch.equal(x, y) for x, y in zip(a["weights"], b["weights"]))
    assert torch.equal(a["exp_avg"], b["exp_avg"]) and torch.equal(a["exp_avg_sq"], b["exp_avg_sq"])
    assert torch.equal(a["mean"], b["mean"]) and torch.equal(a["var"], b["var"]) and a["count"] == b["count"]
    assert torch.equal(a["losses"], b["losses"]) and torch.equal(a["logits"], b["logits"])
    assert float(a["exp_avg"].abs().max()) > 0 and a["count"] == 1.0 + n_steps * 3 * wk.BS
    # (2) both stepped on the same global minibatches; block w of every batch is rank w's contribution
    assert torch.equal(a["batches"], b["batches"]) and a["batches"].shape == (n_steps, 3, wk.BS, wk.C)
    r = wk.BS // world
    for w in range(world):
        expert, rollouts = wk.make_inputs(w)
        for k in range(n_steps):
            rows = rollouts[k // (wk.EPOCHS * wk.MBS)].reshape(-1, wk.C)
            blk = a["batches"][k, :, w * r:(w + 1) * r]
            assert bool((blk[0].unsqueeze(1) == rows.unsqueeze(0)).all(dim=2).any(dim=1).all())      # policy rows: rank w's rollout
            assert bool((blk[2].unsqueeze(1) == expert.unsqueeze(0)).all(dim=2).any(dim=1).all())    # motion rows: rank w's dataset
            if k < wk.EPOCHS * wk.MBS:
                assert torch.equal(blk[1], blk[0])      # first update: empty replay ring -> the policy rows stand in
            else:
                first = rollouts[0].reshape(-1, wk.C)   # later: rank w's replay ring holds its first rollout
                assert bool((blk[1].unsqueeze(1) == first.unsqueeze(0)).all(dim=2).any(dim=1).all())
    # each rank appended only its own rollout rows to its own ring
    for w, rec in enumerate((a, b)):
        _, rollouts = wk.make_inputs(w)
        allrows = torch.cat([x.reshape(-1, wk.C) for x in rollouts])
        assert rec["replay_len"] == allrows.shape[0] == rec["replay_head"]
        assert bool((rec["replay_rows"].unsqueeze(1)[:64] == allrows.unsqueeze(0)).all(dim=2).any(dim=1).all())
    # (3) == a single process stepped on the concatenated rows (the recorded global minibatches), same initial replica
    w0 = odisc.make_weights(wk.C, seed=4)
    disc = AmpDiscriminator([(x.cuda(), y.cuda()) for x, y in w0], "cuda:0")
    trainer = AmpDiscriminatorTrainer(disc, batch_size=wk.BS, defer_refresh=True)
    losses = []
    for k in range(n_steps):
        p, q, m = (t.cuda() for t in a["batches"][k])
        losses.append(trainer.step(p, q, m)["loss"].clone())
    trainer.refresh()
    torch.cuda.synchronize()
    assert all(torch.equal(x.cpu(), y) for x, y in zip((t for pair in trainer.weights() for t in pair), a["weights"]))
    m1, v1, st1 = trainer.adam_state()
    assert st1 == n_steps and torch.equal(m1.cpu(), a["exp_avg"]) and torch.equal(v1.cpu(), a["exp_avg_sq"])
    mean, var, count = trainer.scaler_state()
    assert torch.equal(mean.cpu(), a["mean"]) and torch.equal(var.cpu(), a["var"]) and count == a["count"]
    assert torch.equal(torch.stack(losses).cpu(), a["losses"])


def test_a_group_of_one_rank_equals_the_single_rank_flow():
    """AmpDiscriminatorUpdate(group=<world of one>) == AmpDiscriminatorUpdate(group=None): same batches, losses, weights."""
    import torch.distributed as dist

    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer, AmpDiscriminatorUpdate, AmpReplayBuffer

    C, bs, epochs, mbs = 166, 256, 2, 2
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        w = odisc.make_weights(C, seed=4)
        gen = torch.Generator().manual_seed(1)
        expert = torch.randn(3000, C, generator=gen).cuda()
        rollouts = [torch.randn(8, 128, C, generator=gen).cuda() for _ in range(3)]
        runs = []
        for group in (None, dist.group.WORLD):
            disc = AmpDiscriminator([(a.cuda(), b.cuda()) for a, b in w], "cuda:0")
            trainer = AmpDiscriminatorTrainer(disc, batch_size=bs, defer_refresh=True)
            replay, motion = AmpReplayBuffer(5000, C, "cuda:0", seed=5), AmpReplayBuffer(2500, C, "cuda:0", seed=6)
            motion.add_samples(expert)
            upd = AmpDiscriminatorUpdate(trainer, replay, motion, learning_epochs=epochs, mini_batches=mbs, seed=9, record_batches=True,
                                         prefetch=False, group=group)
            batches, losses = [], []
            for r in rollouts:
                losses += [l.clone() for l in upd.update(r)]
                batches += [torch.stack(b) for b in upd.batches]
            torch.cuda.synchronize()
            runs.append((torch.stack(batches), torch.stack(losses), [t.clone() for pair in trainer.weights() for t in pair],
                         replay.sample(4096), len(replay)))
        (b0, l0, w0, r0, n0), (b1, l1, w1, r1, n1) = runs
        assert torch.equal(b0, b1) and torch.equal(l0, l1) and n0 == n1 and torch.equal(r0, r1)
        assert all(torch.equal(x, y) for x, y in zip(w0, w1))
    finally:
        dist.destroy_process_group()
