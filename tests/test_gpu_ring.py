"""GPU parity: device ring buffers (skrl RandomMemory semantics; skrl absent => parity unpinned) vs oracle/rng.py.
Bit-exact: stored rows, write head, sampled indices and sampled rows."""

import numpy as np
import pytest
import torch

from oracle import rng as orng

pytestmark = pytest.mark.gpu


def test_append_wraps_like_a_ring_and_sampling_is_bit_exact():
    from humanoid_amp_amd.engine import AmpReplayBuffer

    cap, dim = 1000, 166
    buf = AmpReplayBuffer(cap, dim, "cuda:0", seed=77)
    ref = orng.RingOracle(cap, dim)
    g = torch.Generator().manual_seed(5)
    assert len(buf) == 0
    for n in (1, 255, 600, 400, 1, 2500, 37):   # partial fill, exact wrap, a batch larger than the memory
        batch = torch.randn(n, dim, generator=g)
        buf.add_samples(batch.cuda())
        ref.add(batch.numpy())
        assert len(buf) == ref.size and buf.memory_index == ref.head
        for draw_rows in (64, 4096):
            draw = buf._draw
            rows, idx = buf.sample(draw_rows, return_indices=True)
            want = orng.ring_sample_indices(ref.size, 77, draw, draw_rows)
            assert np.array_equal(idx.cpu().numpy(), want)
            assert np.array_equal(rows.cpu().numpy(), ref.rows[want])
    # every slot holds what the oracle holds
    all_rows, idx = buf.sample(20000, return_indices=True)
    assert np.array_equal(all_rows.cpu().numpy(), ref.rows[idx.cpu().numpy()])
    assert len(np.unique(idx.cpu().numpy())) == cap   # 20 draws per slot on average: every slot is hit


def test_strided_input_views_and_errors():
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import AmpReplayBuffer

    buf = AmpReplayBuffer(64, 83, "cuda:0")
    with pytest.raises(nat.AmpEngineError):
        buf.sample(4)                                   # empty
    big = torch.randn(10, 2, 83, device="cuda")
    buf.add_samples(big[:, 0, :])                       # row stride 166: a view, no copy
    rows, idx = buf.sample(32, return_indices=True)
    assert torch.equal(rows, big[:, 0, :][idx])
    with pytest.raises(nat.AmpEngineError):
        buf.add_samples(torch.randn(3, 84, device="cuda"))


def test_uniformity_of_the_draw():
    """Chi-square of 1 M draws over 50 slots stays far inside the 0.1 % tail."""
    from humanoid_amp_amd.engine import AmpReplayBuffer

    buf = AmpReplayBuffer(50, 4, "cuda:0", seed=3)
    buf.add_samples(torch.arange(200, dtype=torch.float32, device="cuda").view(50, 4))
    _, idx = buf.sample(1_000_000, return_indices=True)
    counts = torch.bincount(idx, minlength=50).double().cpu().numpy()
    chi2 = float(((counts - 20000.0) ** 2 / 20000.0).sum())
    assert chi2 < 90.0   # 49 dof: P(chi2 > 85.4) = 0.001


def test_first_row_slices_one_minibatch_draw():
    """A minibatch drawn by W ranks (each its rows [w * r, (w + 1) * r) at first_row = w * r) consumes the variates of ONE
    batch-size draw: the per-rank index lists are the slices of the whole draw, bit for bit, and match oracle/rng.py."""
    from humanoid_amp_amd.engine import AmpReplayBuffer

    buf = AmpReplayBuffer(5000, 166, "cuda:0", seed=11)
    buf.add_samples(torch.randn(3333, 166, generator=torch.Generator().manual_seed(2)).cuda())
    B = 4096
    buf._draw = 7
    rows, idx = buf.sample(B, return_indices=True)
    assert np.array_equal(idx.cpu().numpy(), orng.ring_sample_indices(3333, 11, 7, B))
    for world in (2, 8):
        r = B // world
        for w in range(world):
            buf._draw = 7
            part, pidx = buf.sample(r, return_indices=True, first_row=w * r)
            assert torch.equal(pidx, idx[w * r:(w + 1) * r]) and torch.equal(part, rows[w * r:(w + 1) * r])
            assert np.array_equal(pidx.cpu().numpy(), orng.ring_sample_indices(3333, 11, 7, r, first_row=w * r))
    # counters past 2^32 use the high counter word
    buf._draw = 1
    _, hi = buf.sample(64, return_indices=True, first_row=(1 << 32) + 5)
    assert np.array_equal(hi.cpu().numpy(), orng.ring_sample_indices(3333, 11, 1, 64, first_row=(1 << 32) + 5))
