"""GPU parity: MotionLoader on the HIP engine vs the golden vectors (reference outputs) and the oracle.

Bars: frame indices / blend bit-exact; LERP tables bit-exact (same fp32 ops, contraction off);
SLERP within 1e-5 (device acosf/sinf differ from the host libm by ulps), written in the asserts.
"""

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu

TOL = 1e-5
LERP_TABLES = ("dof_positions", "dof_velocities", "body_positions", "body_linear_velocities", "body_angular_velocities")
TABLE_KEYS = ("dof_positions", "dof_velocities", "body_positions", "body_rotations", "body_linear_velocities",
              "body_angular_velocities")


@pytest.fixture(scope="module")
def loaders():
    from humanoid_amp_amd.motions import MotionLoader

    return {tag: MotionLoader(",".join(gu.clip_files(tag)), "cuda:0") for tag in gu.CLIPSETS}


@pytest.mark.parametrize("tag", list(gu.CLIPSETS))
def test_tables_and_meta(tag, loaders):
    fx = gu.golden(f"frame_blend_{tag}")
    ml = loaders[tag]
    assert float(ml.dt) == float(fx["dt"])
    assert np.array_equal(ml.durations, fx["durations"])
    assert np.array_equal(ml.traj_starts, fx["traj_starts"]) and np.array_equal(ml.traj_ends, fx["traj_ends"])


@pytest.mark.parametrize("tag", list(gu.CLIPSETS))
def test_frame_blend_bit_exact(tag, loaders):
    fx = gu.golden(f"frame_blend_{tag}")
    # numpy in -> numpy out, like the reference's private helper (motion_loader.py:281-307)
    i0, i1, b = loaders[tag]._compute_frame_blend(fx["times"], fx["motion_ids"])
    assert isinstance(i0, np.ndarray) and i0.dtype == np.int64 and i1.dtype == np.int64 and b.dtype == np.float64
    assert np.array_equal(i0, fx["index_0"]) and np.array_equal(i1, fx["index_1"]) and np.array_equal(b, fx["blend"])
    # device tensors in -> device tensors out (no host round trip), same bits
    d0, d1, db = loaders[tag]._compute_frame_blend(torch.from_numpy(fx["times"]).cuda(), torch.from_numpy(fx["motion_ids"]).cuda())
    assert d0.is_cuda and d0.dtype == torch.int64 and db.dtype == torch.float64
    assert np.array_equal(d0.cpu().numpy(), i0) and np.array_equal(d1.cpu().numpy(), i1) and np.array_equal(db.cpu().numpy(), b)


@pytest.mark.parametrize("tag", list(gu.CLIPSETS))
def test_sample_vs_reference(tag, loaders):
    fx = gu.golden(f"sample_{tag}")
    outs = loaders[tag].sample(len(fx["times"]), times=fx["times"], motion_ids=fx["motion_ids"])
    for name, o in zip(TABLE_KEYS, outs):
        got, want = o.cpu().numpy(), fx[name]
        assert got.shape == want.shape and got.dtype == np.float32
        if name in LERP_TABLES:
            assert np.array_equal(got, want), name  # bit-exact
        else:
            assert np.nanmax(np.abs(got - want)) <= TOL, name
            assert np.array_equal(np.isnan(got), np.isnan(want))
    # device-tensor inputs take the no-copy path and must agree with the numpy path
    t = torch.from_numpy(fx["times"]).cuda()
    ids = torch.from_numpy(fx["motion_ids"]).cuda()
    outs2 = loaders[tag].sample(len(fx["times"]), times=t, motion_ids=ids)
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("tag", list(gu.CLIPSETS))
def test_sample_default_ids(tag, loaders):
    fx = gu.golden(f"sample_defaultids_{tag}")
    outs = loaders[tag].sample(len(fx["times"]), times=fx["times"])
    for name, o in zip(TABLE_KEYS, outs):
        got = o.cpu().numpy()
        if name in LERP_TABLES:
            assert np.array_equal(got, fx[name]), name
        else:
            assert np.max(np.abs(got - fx[name])) <= TOL, name


def test_sample_on_frame_returns_table_rows(loaders):
    ml = loaders["g1_walk"]
    k = np.array([0, 1, 17, 200, 398])
    outs = ml.sample(5, times=k * ml.dt, motion_ids=np.zeros(5, dtype=np.int64))
    assert torch.equal(outs[0], ml.dof_positions[k])
    assert torch.equal(outs[2], ml.body_positions[k])
    assert torch.equal(outs[4], ml.body_linear_velocities[k])


def test_sample_empty_and_ragged(loaders):
    ml = loaders["humanoid3"]
    outs = ml.sample(0, times=np.zeros(0), motion_ids=np.zeros(0, dtype=np.int64))
    assert [o.shape[0] for o in outs] == [0] * 6
    with pytest.raises(IndexError):
        ml.sample(2, times=np.zeros(2), motion_ids=np.array([0, 3]))
    with pytest.raises(ValueError):
        ml.sample(2, times=np.zeros(2), motion_ids=np.array([0, 1, 2]))
    # a tile boundary (256 samples per workgroup) and a ragged tail
    from oracle import motion as om

    mt = om.load_tables(gu.clip_files("humanoid3"))
    rng = np.random.default_rng(5)
    n = 256 * 3 + 37
    ids = rng.integers(0, 3, size=n)
    t = rng.uniform(-0.05, 1.02, size=n) * mt.durations[ids]
    want = om.sample(mt, t, ids)
    got = ml.sample(n, times=t, motion_ids=ids)
    for name, g, w in zip(TABLE_KEYS, got, want):
        if name in LERP_TABLES:
            assert torch.equal(g.cpu(), w), name
        else:
            assert float((g.cpu() - w).abs().max()) <= TOL, name


@pytest.mark.parametrize("tag", list(gu.CLIPSETS))
def test_sample_tiles_vs_oracle(tag, loaders):
    """Round 3's sample kernel (padded table, a lane owns a column quad and walks the tile's samples, outputs staged in LDS):
    64-sample tiles for G1_walk / the humanoid clips, 16-sample tiles for G1_dance (39 bodies); several whole tiles + a
    ragged tail against the oracle, LERP tables bit-exact."""
    from oracle import motion as om

    ml = loaders[tag]
    mt = om.load_tables(gu.clip_files(tag))
    rng = np.random.default_rng(11)
    n = 64 * 21 + 13
    ids = rng.integers(0, ml.num_trajectories, size=n)
    t = rng.uniform(-0.1, 1.03, size=n) * mt.durations[ids]
    want = om.sample(mt, t, ids)
    got = ml.sample(n, times=t, motion_ids=ids)
    for name, g, w in zip(TABLE_KEYS, got, want):
        if name in LERP_TABLES:
            assert torch.equal(g.cpu(), w), name
        else:
            d = (g.cpu() - w).abs()
            assert float(d[~torch.isnan(d)].max()) <= TOL and torch.equal(torch.isnan(g.cpu()), torch.isnan(w)), name


@pytest.mark.parametrize("tag", list(gu.CLIPSETS))
def test_sample_large_batch_equals_its_chunks(tag, loaders):
    """One large ragged batch (33 545 samples) against the same samples in chunks of 5 000, bit for bit, NaN patterns included
    -- and the first 2 000 samples against the oracle; with four outputs NULL the remaining two do not change.  (Written for the
    LDS-resident-table variant of the kernel tried in round 3 -- profiles/r03_sample_kernel.md -- and kept: whatever kernel a batch
    size selects, a sample's result may not depend on its neighbours.)"""
    import ctypes as C

    from humanoid_amp_amd import _native as nat
    from oracle import motion as om

    ml = loaders[tag]
    mt = om.load_tables(gu.clip_files(tag))
    rng = np.random.default_rng(23)
    n = 16384 * 2 + 777
    ids = rng.integers(0, ml.num_trajectories, size=n)
    t = rng.uniform(-0.1, 1.03, size=n) * mt.durations[ids]
    big = ml.sample(n, times=t, motion_ids=ids)
    for a in range(0, n, 5000):
        part = ml.sample(min(5000, n - a), times=t[a:a + 5000], motion_ids=ids[a:a + 5000])
        for name, g, w in zip(TABLE_KEYS, big, part):
            gs = g[a:a + 5000]
            assert torch.equal(torch.isnan(gs), torch.isnan(w)) and torch.equal(torch.nan_to_num(gs), torch.nan_to_num(w)), (name, a)
    want = om.sample(mt, t[:2000], ids[:2000])
    for name, g, w in zip(TABLE_KEYS, big, want):
        if name in LERP_TABLES:
            assert torch.equal(g[:2000].cpu(), w), name
    # NULL outputs: only body_positions and body_rotations asked for
    td, idd = torch.from_numpy(t).cuda(), torch.from_numpy(ids).cuda()
    bp, br = torch.full_like(big[2], -7.0), torch.full_like(big[3], -7.0)
    null = C.c_void_p(None)
    with torch.cuda.device("cuda:0"):
        nat.check(nat.load().amp_motion_sample(ml._need_handle(), nat.dptr(td), nat.dptr(idd), n, null, null, nat.dptr(bp), nat.dptr(br), null,
                                               null, nat.stream_ptr()), "amp_motion_sample")
    assert torch.equal(torch.nan_to_num(bp), torch.nan_to_num(big[2])) and torch.equal(torch.nan_to_num(br), torch.nan_to_num(big[3]))


def test_sample_skips_null_outputs(loaders):
    """amp_motion_sample: any output pointer may be NULL (include/amp_engine.h); the others are unchanged by that."""
    import ctypes as C

    from humanoid_amp_amd import _native as nat

    ml = loaders["g1_walk"]
    n = 200
    rng = np.random.default_rng(3)
    t = torch.from_numpy(rng.uniform(0, 1, size=n) * ml.durations[0]).cuda()
    full = ml.sample(n, times=t)
    bp = torch.full_like(full[2], -7.0)
    br = torch.full_like(full[3], -7.0)
    null = C.c_void_p(None)
    with torch.cuda.device("cuda:0"):
        nat.check(nat.load().amp_motion_sample(ml._need_handle(), nat.dptr(t), null, n, null, null, nat.dptr(bp), nat.dptr(br), null, null,
                                               nat.stream_ptr()), "amp_motion_sample")
    assert torch.equal(bp, full[2]) and torch.equal(br, full[3])


@pytest.mark.parametrize("name,tag,keys", [("g1_walk_k2", "g1_walk", gu.G1_KEY_BODIES), ("g1_walk_k10", "g1_walk", gu.G1_KEY_BODIES),
                                           ("g1_dance_k10", "g1_dance", gu.G1_KEY_BODIES),
                                           ("humanoid3_k2", "humanoid3", gu.HUM_KEY_BODIES)])
def test_collect_reference_vs_reference(name, tag, keys, loaders):
    fx = gu.golden(f"collect_{name}")
    ml = loaders[tag]
    D = ml.set_obs_layout(fx["motion_dof_indexes"].tolist(), int(fx["motion_ref_body_index"]),
                          fx["motion_key_body_indexes"].tolist())
    K = int(fx["num_amp_observations"])
    out = ml.collect_reference(fx["times"], fx["motion_ids"], K)
    assert out.shape == (len(fx["times"]), K * D)
    got, want = out.cpu().numpy(), fx["amp_obs"]
    nd2 = 2 * ml.num_dofs
    cols = np.arange(K * D) % D
    slerp_cols = (cols > nd2) & (cols <= nd2 + 6)  # tangent | normal: the only SLERP-derived features
    assert np.array_equal(got[:, ~slerp_cols], want[:, ~slerp_cols])  # LERP features bit-exact
    assert np.max(np.abs(got - want)) <= TOL
    # scatter form: out[dst_rows] = expert rows (reset path)
    buf = torch.zeros((100, K, D), device="cuda")
    rows = torch.from_numpy(np.random.default_rng(0).permutation(100)[: len(fx["times"])].astype(np.int64)).cuda()
    ml.collect_reference(fx["times"], fx["motion_ids"], K, out=buf, dst_rows=rows)
    assert torch.equal(buf[rows].view(len(fx["times"]), -1), out)
    untouched = torch.ones(100, dtype=torch.bool, device="cuda")
    untouched[rows] = False
    assert float(buf[untouched].abs().max()) == 0.0


@pytest.mark.parametrize("n,K", [(1, 1), (31, 2), (129, 2), (1000, 3), (27, 10), (333, 10)])
def test_collect_reference_wide_tiles_match_the_scatter_form(n, K, loaders):
    """The contiguous form runs 256-sample workgroups (a lane owns a column quad and walks samples), the scatter form the
    64-sample body: the same rows bit for bit over ragged sizes, several K and every clip set."""
    for tag, keys in (("g1_walk", gu.G1_KEY_BODIES), ("g1_dance", gu.G1_KEY_BODIES), ("humanoid3", gu.HUM_KEY_BODIES)):
        ml = loaders[tag]
        fx = gu.golden(f"collect_{'g1_walk_k2' if tag == 'g1_walk' else 'g1_dance_k10' if tag == 'g1_dance' else 'humanoid3_k2'}")
        D = ml.set_obs_layout(fx["motion_dof_indexes"].tolist(), int(fx["motion_ref_body_index"]), fx["motion_key_body_indexes"].tolist())
        rng = np.random.default_rng(n * 31 + K)
        ids = rng.integers(0, ml.num_trajectories, size=n)
        t = rng.uniform(-0.1, 1.05, size=n) * ml.durations[ids]
        wide = ml.collect_reference(t, ids, K)
        buf = torch.zeros((n + 5, K, D), device="cuda")
        rows = torch.from_numpy(rng.permutation(n + 5)[:n].astype(np.int64)).cuda()
        ml.collect_reference(t, ids, K, out=buf, dst_rows=rows)
        assert torch.equal(buf[rows].view(n, -1), wide), (tag, n, K)
