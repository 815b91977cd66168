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


def test_command_draw_distribution_determinism_and_domain_separation():
    """a14 (g1_amp_env.py:146-167,421-439): the engine's counter-based command draw.  Ranges g1_amp_env_cfg.py:96-100."""
    env = np.arange(200000)
    c, t = rng.command_draw(5, 3, env, 0, -1.0, 2.0, 4.0, 3.0)
    assert c.dtype == np.float32 and t.dtype == np.float32 and c.shape == (200000, 2)
    assert -1.0 <= c.min() and c.max() < 1.0 and 4.0 <= t.min() and t.max() < 7.0
    assert abs(c.mean()) < 5e-3 and abs(c.var() - 4 / 12) < 5e-3 and abs(t.mean() - 5.5) < 1e-2 and abs(t.var() - 9 / 12) < 1e-2
    assert abs(np.corrcoef(c[:, 0], c[:, 1])[0, 1]) < 1e-2 and abs(np.corrcoef(c[:, 0], t)[0, 1]) < 1e-2
    c2, t2 = rng.command_draw(5, 3, env[::-1].copy(), 0, -1.0, 2.0, 4.0, 3.0)
    assert np.array_equal(c, c2[::-1]) and np.array_equal(t, t2[::-1])  # depends on (seed, step, env, mode) only
    c3, _ = rng.command_draw(5, 3, env, 1, -1.0, 2.0, 4.0, 3.0)         # the reset draw of the same env / step differs
    assert (c3 != c).mean() > 0.99
    # ... and neither reuses the words of the reset-time draw of the same (seed, step, env)
    r = rng.philox4x32_10(env, 0, 3, 0, 5, 0)
    u0 = (r[0] >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24) * np.float32(2.0) + np.float32(-1.0)
    assert (u0 != c[:, 0]).mean() > 0.99


def test_command_tick_and_reset_semantics():
    cmd = np.zeros((6, 2), np.float32)
    left = np.array([0.5, 0.02, 0.0, -1.0, np.inf, 0.033333], np.float32)
    c, t = rng.command_tick(cmd, left, 1 / 30, (-1.0, 1.0), (4.0, 7.0), seed=1, step=9, env_offset=100)
    assert np.allclose(t[0], 0.5 - np.float32(1 / 30)) and t[4] == np.inf and not c[0].any() and not c[4].any()
    for e in (1, 2, 3):  # expired: resampled with the GLOBAL env id
        ce, te = rng.command_draw(1, 9, np.array([100 + e]), 0, -1.0, 2.0, 4.0, 3.0)
        assert np.array_equal(c[e], ce[0]) and t[e] == te[0] and 4.0 <= t[e] < 7.0
    # empty range: timers run down, nothing is resampled (g1_amp_env.py:151-154)
    c, t = rng.command_tick(cmd, left, 1 / 30, (0.5, 0.5), (4.0, 7.0), seed=1, step=9)
    assert not c.any() and t[3] < -1.0
    # reset side: fixed command + infinite timer when the range is empty (g1_amp_env.py:436-439)
    c, t = rng.command_reset(cmd, left, np.array([1, 4]), (0.5, 0.5), (4.0, 7.0), seed=1, step=9)
    assert c[1].tolist() == [0.5, 0.0] and c[4].tolist() == [0.5, 0.0] and np.isinf(t[[1, 4]]).all() and not c[0].any()
