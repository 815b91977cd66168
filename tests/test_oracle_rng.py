"""CPU: the Philox oracle against Random123's published known-answer vectors (kat_vectors, philox4x32-10)."""

import numpy as np

from oracle import rng


def _one(ctr, key):
    out = rng.philox4x32_10(*[np.array([c], dtype=np.uint64) for c in ctr], *key)
    return [int(o[0]) for o in out]


def test_philox4x32_10_known_answers():
    assert _one((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert _one((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert _one((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_sample_times_distribution_and_determinism():
    dur = np.array([2.55, 1.35, 15.0166666667])
    idx = np.arange(200000)
    ids, t = rng.sample_times(dur, seed=42, step=7, index=idx)
    ids2, t2 = rng.sample_times(dur, seed=42, step=7, index=idx[::-1].copy())
    assert np.array_equal(ids, ids2[::-1]) and np.array_equal(t, t2[::-1])  # a draw depends on (seed, step, index) only
    assert ids.min() == 0 and ids.max() == 2 and abs(np.bincount(ids) / len(ids) - 1 / 3).max() < 5e-3
    u = t / dur[ids]
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 3e-3 and abs(u.var() - 1 / 12) < 2e-3
    ids3, t3 = rng.sample_times(dur, seed=42, step=8, index=idx)
    assert (ids3 != ids).mean() > 0.5 and not t3[:100].tolist() == t[:100].tolist()
    assert not rng.sample_times(dur, 1, 1, idx[:10], start=True)[1].any()
