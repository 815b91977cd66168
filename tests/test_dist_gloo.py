"""CPU, world_size 2, gloo: the N > 1 plumbing -- env sharding needs no data-path collective (concatenated shard
outputs == single-shard output) and the AMP replay minibatch all-gather is rank-major and complete."""

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
    from humanoid_amp_amd.distributed import ReplayAllGather, allgather_minibatch, shard_bounds
    from oracle import env as oenv
    from oracle import motion as om

    # (1) all-gather: rank-major, nothing lost, no staging copy needed
    shard = torch.full((5, 7), float(rank)) + torch.arange(5).float()[:, None] * 0.01
    full = allgather_minibatch(shard)
    assert full.shape == (world * 5, 7)
    for r in range(world):
        assert torch.equal(full[r * 5:(r + 1) * 5], torch.full((5, 7), float(r)) + torch.arange(5).float()[:, None] * 0.01)
    table = torch.arange(40 * 6, dtype=torch.float32).view(40, 6) + 1000 * rank
    rg = ReplayAllGather(table, rows=8, seed=rank)
    got = rg()
    assert got.shape == (world * 8, 6)
    for r in range(world):  # every block comes from rank r's table
        assert bool(((got[r * 8:(r + 1) * 8] // 1000).long() == r).all())
    rg3 = ReplayAllGather(table, rows=4, seed=rank, slots=2)  # async slots, wrap-around
    for _ in range(5):
        s = rg3.start()
    rg3.wait_all()
    last = rg3.result(s)
    assert last.shape == (world * 4, 6) and bool(((last[:4] // 1000).long() == 0).all()) and bool(((last[4:] // 1000).long() == 1).all())

    rg4 = ReplayAllGather(table, rows=4, seed=rank, minibatches=3)  # three minibatches fused into one collective
    slot = rg4.start()
    rg4.wait_all()
    assert rg4.result(slot).shape == (world * 12, 6)
    for i in range(3):
        blocks = rg4.minibatch_blocks(slot, i)
        assert len(blocks) == world and all(b.shape == (4, 6) and b.is_contiguous() for b in blocks)
        for r, b in enumerate(blocks):  # block r holds rank r's rows
            assert bool(((b // 1000).long() == r).all())
    # the fused draw is the same row sequence as one long draw of the unfused object
    ref = ReplayAllGather(table, rows=12, seed=rank)
    assert torch.equal(ref(), rg4.result(slot))

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
