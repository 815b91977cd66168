"""CPU, world_size 2, gloo: the N > 1 plumbing -- env sharding needs no data-path collective (concatenated shard
outputs == single-shard output), the all-gather is rank-major and complete, and the discriminator update's exchange
(distributed.UpdateExchange + engine.AmpDiscriminatorUpdate(group=...)) puts every row where the documented mapping says.
The oracle (oracle/rng.py) stands in for the device ring draw here: this is host logic, the HIP path of the same flow is
tests/test_gpu_dist_update.py."""

import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanoid_amp_amd.distributed import UpdateExchange, allgather_minibatch, shard_bounds
    from humanoid_amp_amd.engine import AmpDiscriminatorUpdate
    from oracle import env as oenv
    from oracle import motion as om
    from oracle import rng as orng

    # (1) all-gather: rank-major, nothing lost, no staging copy needed
    shard = torch.full((5, 7), float(rank)) + torch.arange(5).float()[:, None] * 0.01
    full = allgather_minibatch(shard)
    assert full.shape == (world * 5, 7)
    for r in range(world):
        assert torch.equal(full[r * 5:(r + 1) * 5], torch.full((5, 7), float(r)) + torch.arange(5).float()[:, None] * 0.01)

    # (1b) the update's exchange: [steps, groups, r, C] per rank -> [steps, groups, world * r, C], identical on every rank, row j
    # of a minibatch from rank j // r; ONE collective for the whole update
    S, G, r, C = 4, 3, 6, 5
    ex = UpdateExchange(S, G, r, C, "cpu", group=dist.group.WORLD)
    assert (ex.world, ex.rank, ex.first_row) == (world, rank, rank * r) and ex.bytes_per_rank == S * G * r * C * 4
    code = lambda w: (1000.0 * w + 100.0 * torch.arange(S)[:, None, None, None] + 10.0 * torch.arange(G)[None, :, None, None]  # noqa: E731
                      + torch.arange(r)[None, None, :, None] + 0.001 * torch.arange(C)[None, None, None, :])
    ex.contrib.copy_(code(rank))
    ex.start()
    got = ex.finish()
    assert got.shape == (S, G, world * r, C)
    for w in range(world):
        assert torch.equal(got[:, :, w * r:(w + 1) * r], code(w))
    assert ex.source_of(r + 2) == (1, 2)
    both = [torch.empty_like(got) for _ in range(world)]
    dist.all_gather(both, got)
    assert all(torch.equal(b, got) for b in both)

    # (1c) AmpDiscriminatorUpdate(group=...) on test doubles (CPU rings drawing with the engine's counter-based draw as restated in
    # oracle/rng.py, a trainer that records what it is stepped on): which row reaches which (rank, step, group, position)
    class CpuRing:
        def __init__(self, rows, seed):
            self.rows, self.seed, self._draw, self.appended, self.drawn = rows, seed, 0, [], []

        def __len__(self):
            return self.rows.shape[0]

        def sample(self, n, *, out, first_row=0):
            idx = orng.ring_sample_indices(len(self), self.seed, self._draw, n, first_row)
            self.drawn.append((self._draw, first_row, idx))
            self._draw += 1
            out.copy_(self.rows[torch.from_numpy(idx)])

        def add_samples(self, rows):
            self.appended.append(rows.clone())

    def cpu_take(rows, seed, epoch, first, count, out=None):
        """The engine's epoch shuffle (amp_rows_take_permuted) on CPU tensors through its restatement in oracle/rng.py."""
        idx = torch.from_numpy(orng.feistel_permutation(rows.shape[0], seed, epoch, np.arange(first, first + count)))
        res = rows[idx]
        return res if out is None else out.copy_(res)

    class RecordingTrainer:
        batch_size, device, defer_refresh = 8, torch.device("cpu"), False

        def __init__(self):
            self.steps = []

        def step(self, policy, replay, motion):
            self.steps.append(torch.stack([policy, replay, motion]).clone())
            return {"loss": torch.zeros(())}

    cols = 4
    tag = lambda kind, n: torch.stack([torch.full((n,), float(rank)), torch.full((n,), float(kind)), torch.arange(n).float(),  # noqa: E731
                                       torch.zeros(n)], dim=1)
    rollout = tag(0, 40)                                     # [rank, group tag, local row, 0]
    replay, motion = CpuRing(tag(1, 30 + 7 * rank), seed=5), CpuRing(tag(2, 50), seed=6)   # replay rings of different sizes
    trainer = RecordingTrainer()
    upd = AmpDiscriminatorUpdate(trainer, replay, motion, learning_epochs=3, mini_batches=2, seed=9 + rank, group=dist.group.WORLD,
                                 take_rows=cpu_take)
    losses = upd.update(rollout.view(5, 8, cols))
    bs, rr = trainer.batch_size, trainer.batch_size // world
    assert len(losses) == len(trainer.steps) == 6 and upd.exchange.rows_per_rank == rr
    mine = torch.stack(trainer.steps)                        # [6, 3, bs, cols]
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    assert all(torch.equal(e, mine) for e in everyone)       # every replica steps on the same global minibatches
    for k in range(6):
        for g in range(3):
            blk = mine[k, g]
            assert torch.equal(blk[:, 0], torch.arange(world).repeat_interleave(rr).float())   # rows w*r .. (w+1)*r-1 from rank w
            assert bool((blk[:, 1] == g).all())                                                # ... and from that rank's group-g source
        # this rank's block: the ring rows the engine's draw selects at first_row = rank * r (draw counter k)
        for g, ring in ((1, replay), (2, motion)):
            d, first, idx = ring.drawn[k]
            assert (d, first) == (k, rank * rr)
            assert torch.equal(mine[k, g, rank * rr:(rank + 1) * rr, 2], torch.from_numpy(idx).float())
        # policy rows: positions [mb * per, mb * per + rr) of this rank's epoch permutation (per = 40 rows / 2 minibatches)
        want = orng.feistel_permutation(40, 9 + rank, k // 2, np.arange((k % 2) * 20, (k % 2) * 20 + rr))
        assert torch.equal(mine[k, 0, rank * rr:(rank + 1) * rr, 2], torch.from_numpy(want).float())
        assert len(set(want.tolist())) == rr
    # an epoch's two minibatches take disjoint rollout rows
    for e in range(3):
        a, b = (set(mine[2 * e + i, 0, rank * rr:(rank + 1) * rr, 2].tolist()) for i in range(2))
        assert not (a & b)
    # the variates of a minibatch do not depend on the world size: the per-rank draws are slices of ONE batch-size draw
    whole = orng.ring_sample_indices(50, 6, 3, bs)
    assert all((orng.ring_sample_indices(50, 6, 3, rr, first_row=w * rr) == whole[w * rr:(w + 1) * rr]).all() for w in range(world))
    assert len(replay.appended) == 1 and torch.equal(replay.appended[0], rollout)   # the rank's own rollout rows, once

    # (2) env sharding: each rank computes its contiguous env block of the oracle path; rank 0 checks the concatenation
    clips = [os.path.join(ROOT, "humanoid_amp_amd", "motions", "G1_walk.npz")]
    mt = om.load_tables(clips)
    N = 50
    rng = np.random.default_rng(0)
    t = rng.uniform(0, 1, N) * mt.durations[0]
    ids = np.zeros(N, dtype=np.int64)
    lo, hi = shard_bounds(N, world, rank)
    perm, keys = list(range(29)), [7, 8, 9, 10]
    mine = oenv.collect_reference(mt, t[lo:hi], ids[lo:hi], 2, perm, 0, keys)
    parts = [torch.zeros(shard_bounds(N, world, r)[1] - shard_bounds(N, world, r)[0], mine.shape[1]) for r in range(world)]
    dist.all_gather(parts, mine) if hi - lo == parts[0].shape[0] == parts[-1].shape[0] else None
    if rank == 0:
        whole = oenv.collect_reference(mt, t, ids, 2, perm, 0, keys)
        assert torch.equal(torch.cat(parts), whole)  # bit-for-bit: every env is independent
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
