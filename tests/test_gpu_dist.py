"""GPU: the RCCL path of the replay-minibatch all-gather on the one visible device (world size 1: checks that the
`nccl` (= RCCL) backend initialises on this ROCm stack and that all_gather_into_tensor runs on HIP tensors).  The
world-size-2 logic is covered on CPU with gloo (tests/test_dist_gloo.py); 2..8-GPU runs belong to the driver."""

import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_world1_allgather_and_bench_collective():
    from humanoid_amp_amd.distributed import ReplayAllGather, allgather_minibatch

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        shard = torch.randn(4096, 166, device="cuda")
        full = allgather_minibatch(shard, force_collective=True)  # really goes through RCCL
        torch.cuda.synchronize()
        assert torch.equal(full, shard)
        table = torch.randn(20000, 166, device="cuda")
        rg = ReplayAllGather(table, rows=4096, seed=0)
        out = rg()
        assert out.shape == (4096, 166)
        # every gathered row is a row of the table
        assert bool((out[:8].unsqueeze(1) == table.unsqueeze(0)).all(dim=2).any(dim=1).all())
        # asynchronous slots (the bench's schedule)
        rg2 = ReplayAllGather(table, rows=512, seed=1, slots=3)
        rg2._async = True  # world size 1 would short-circuit: force the real async collective
        slots = [rg2.start() for _ in range(5)]  # wraps around: slots 0 and 1 are waited and reused
        rg2.wait_all()
        assert slots == [0, 1, 2, 0, 1] and all(w is None for w in rg2.works)
        assert bool((rg2.result(2)[:4].unsqueeze(1) == table.unsqueeze(0)).all(dim=2).any(dim=1).all())
        t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the max-over-ranks timing reduction of bench.py
        dist.barrier()
        assert float(t) == 1.5
    finally:
        dist.destroy_process_group()
