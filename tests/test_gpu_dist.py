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


def test_rccl_world1_allgather_and_update_exchange():
    from humanoid_amp_amd.distributed import UpdateExchange, allgather_minibatch

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        shard = torch.randn(4096, 166, device="cuda")
        full = allgather_minibatch(shard, force_collective=True)  # really goes through RCCL
        torch.cuda.synchronize()
        assert torch.equal(full, shard)
        # the update's exchange at BASELINE's shape (12 steps x 3 groups x 4096 rows x 166 floats = 97.9 MB), forced through the
        # asynchronous RCCL collective + the strided re-blocking copy although the world is one rank
        ex = UpdateExchange(12, 3, 4096, 166, "cuda", group=dist.group.WORLD, force_collective=True)
        assert ex.batches.data_ptr() != ex.contrib.data_ptr() and ex.bytes_per_rank == 12 * 3 * 4096 * 166 * 4
        for trial in range(2):                       # the buffers are reused from update to update
            ex.contrib.normal_()
            ex.start()
            got = ex.finish()
            assert got.shape == (12, 3, 4096, 166) and torch.equal(got, ex.contrib)
        # without the flag a world of one rank is a pass-through (no copy)
        ex1 = UpdateExchange(2, 3, 64, 166, "cuda", group=dist.group.WORLD)
        ex1.start()
        assert ex1.finish().data_ptr() == ex1.contrib.data_ptr()
        t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the max-over-ranks timing reduction of bench.py
        dist.barrier()
        assert float(t) == 1.5
    finally:
        dist.destroy_process_group()
