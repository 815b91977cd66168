"""GPU: the discriminator-update data flow (rollout minibatches x replay / motion ring draws -> training steps -> replay
append) against a numpy restatement of the same flow.  skrl absent: parity unpinned.  The arithmetic of a training step
is covered by tests/test_gpu_disc_train.py; here the ROWS that reach it and the ring contents are checked bit for bit."""

import numpy as np
import pytest
import torch

from oracle import disc as odisc
from oracle import rng as orng

pytestmark = pytest.mark.gpu


def test_update_feeds_the_trainer_what_the_oracle_flow_selects():
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer, AmpDiscriminatorUpdate, AmpReplayBuffer

    C, bs, epochs, mbs = 166, 256, 3, 2
    g = torch.Generator().manual_seed(0)
    w = odisc.make_weights(C, seed=2)
    disc = AmpDiscriminator([(a.cuda(), b.cuda()) for a, b in w], "cuda:0")
    trainer = AmpDiscriminatorTrainer(disc, batch_size=bs)
    replay, motion = AmpReplayBuffer(3000, C, "cuda:0", seed=5), AmpReplayBuffer(2000, C, "cuda:0", seed=6)
    o_replay, o_motion = orng.RingOracle(3000, C), orng.RingOracle(2000, C)
    expert = torch.randn(2500, C, generator=g)            # more than the dataset holds: it wraps
    motion.add_samples(expert.cuda())
    o_motion.add(expert.numpy())
    upd = AmpDiscriminatorUpdate(trainer, replay, motion, learning_epochs=epochs, mini_batches=mbs, seed=9, record_batches=True)
    w_before = [t.clone() for pair in trainer.weights() for t in pair]
    for it in range(3):                                    # 1st update: empty replay buffer -> the policy batch stands in
        rollout = torch.randn(16, 64, C, generator=g)      # [rollouts, envs, K*D]
        rep_draw0, mot_draw0 = replay._draw, motion._draw
        losses = upd.update(rollout.cuda())
        assert len(losses) == epochs * mbs and all(bool(torch.isfinite(l).all()) for l in losses)
        rows = rollout.reshape(-1, C).numpy()
        for i, (pol, rep, mot) in enumerate(upd.batches):
            assert pol.shape == rep.shape == mot.shape == (bs, C)
            want_mot = o_motion.rows[orng.ring_sample_indices(o_motion.size, 6, mot_draw0 + i, bs)]
            assert np.array_equal(mot.cpu().numpy(), want_mot)
            if o_replay.size == 0:
                assert torch.equal(rep, pol)
            else:
                want_rep = o_replay.rows[orng.ring_sample_indices(o_replay.size, 5, rep_draw0 + i, bs)]
                assert np.array_equal(rep.cpu().numpy(), want_rep)
            # the policy batch: positions [mb * per, mb * per + bs) of the epoch's permutation of this rollout's rows (bs distinct rows)
            epoch, mb = it * epochs + i // mbs, i % mbs
            per = rows.shape[0] // mbs
            sel = orng.feistel_permutation(rows.shape[0], 9, epoch, np.arange(mb * per, mb * per + bs))
            assert np.array_equal(pol.cpu().numpy(), rows[sel]) and len(set(sel.tolist())) == bs
        o_replay.add(rows)
        assert len(replay) == o_replay.size and replay.memory_index == o_replay.head
        got, idx = replay.sample(4096, return_indices=True)
        assert np.array_equal(got.cpu().numpy(), o_replay.rows[idx.cpu().numpy()])
    w_after = [t for pair in trainer.weights() for t in pair]
    assert any(not torch.equal(a, b) for a, b in zip(w_before, w_after))   # Adam moved the weights
    # the inference engine sees the trained weights (planes refreshed): forward == oracle forward of the new weights
    x = torch.randn(512, C, generator=g)
    mean, var, _ = trainer.scaler_state()
    ref = odisc.forward([(w_after[0].cpu(), w_after[1].cpu()), (w_after[2].cpu(), w_after[3].cpu()), (w_after[4].cpu(), w_after[5].cpu())],
                        x, mean.cpu(), var.cpu())
    disc.set_scaler(mean, var)
    out = disc.style_reward(x.cuda(), want_logits=True)
    assert float((out["logits"].cpu() - ref["logits"]).abs().max()) <= 1e-5


def test_prefetched_update_equals_the_in_line_flow():
    """AmpDiscriminatorUpdate(prefetch=True) produces the batches of step k + 1 on a side stream under step k, into two
    alternating static buffer sets: over three updates (empty replay buffer first, then wrapped draws) every batch of every
    step, every loss, the trained weights and the replay ring are bit-identical to the in-line flow."""
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer, AmpDiscriminatorUpdate, AmpReplayBuffer

    C, bs, epochs, mbs = 166, 512, 3, 2
    w = odisc.make_weights(C, seed=4)
    gen = torch.Generator().manual_seed(1)
    expert = torch.randn(3000, C, generator=gen).cuda()
    rollouts = [torch.randn(8, 256, C, generator=gen).cuda() for _ in range(3)]
    runs = []
    for prefetch in (False, True):
        disc = AmpDiscriminator([(a.cuda(), b.cuda()) for a, b in w], "cuda:0")
        trainer = AmpDiscriminatorTrainer(disc, batch_size=bs, defer_refresh=True)
        replay, motion = AmpReplayBuffer(5000, C, "cuda:0", seed=5), AmpReplayBuffer(2500, C, "cuda:0", seed=6)
        motion.add_samples(expert)
        upd = AmpDiscriminatorUpdate(trainer, replay, motion, learning_epochs=epochs, mini_batches=mbs, seed=9, record_batches=True,
                                     prefetch=prefetch)
        batches, losses = [], []
        for r in rollouts:
            losses += [l.clone() for l in upd.update(r)]
            batches += upd.batches
        torch.cuda.synchronize()
        runs.append((batches, losses, [t.clone() for pair in trainer.weights() for t in pair], replay.sample(4096), len(replay)))
    (b0, l0, w0, r0, n0), (b1, l1, w1, r1, n1) = runs
    assert len(b0) == len(b1) == 3 * epochs * mbs and n0 == n1
    for (p0, q0, m0), (p1, q1, m1) in zip(b0, b1):
        assert torch.equal(p0, p1) and torch.equal(q0, q1) and torch.equal(m0, m1)
    assert all(torch.equal(a, b) for a, b in zip(l0, l1))
    assert all(torch.equal(a, b) for a, b in zip(w0, w1))
    assert torch.equal(r0, r1)
