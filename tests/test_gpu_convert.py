"""GPU parity: CSV -> npz converter (csrc/convert.hip) vs the oracle and vs the reference's own shipped clips
(see tests/test_oracle_convert.py for what those files are)."""

import os

import numpy as np
import pytest

from oracle import convert as oc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("fixture,gen,ang_tol", [("convert_g1_walk", 2, 1e-5), ("convert_custom_motion", 1, 2.5e-2)])
def test_converter_reproduces_the_shipped_clip(fixture, gen, ang_tol, tmp_path):
    from humanoid_amp_amd.motions.convert import G1_MODEL, MotionConverter, save_motion_npz
    from humanoid_amp_amd.motions import MotionLoader

    g = np.load(os.path.join(HERE, "golden", fixture + ".npz"))
    joints, bodies = [str(n) for n in g["dof_names"]], [str(n) for n in g["body_names"]]
    conv = MotionConverter(G1_MODEL, joints, bodies, "cuda:0")
    out = conv.convert(g["csv_rows"], fps=int(g["fps"]), numpy_generation=gen)
    ref = oc.convert(g["csv_rows"], oc.load_model(G1_MODEL), joints, bodies, fps=int(g["fps"]), promotion=f"numpy{gen}")
    for k in ("dof_positions", "dof_velocities", "body_positions", "body_rotations", "body_linear_velocities", "body_angular_velocities"):
        assert out[k].dtype == g[k].dtype and out[k].shape == g[k].shape, k
    # --- against the reference's file.  Measured on MI355X (tests/perf/convert_bench.py): every array below is BIT-IDENTICAL
    # to the shipped clip; the bars leave one float32 ulp for the FK-derived ones (device sin / cos are not libm's)
    assert np.array_equal(out["dof_positions"], g["dof_positions"])              # float64, scipy's interp1d arithmetic
    assert np.array_equal(out["dof_velocities"], g["dof_velocities"])            # float64, differences + scipy's Gaussian
    assert np.abs(out["body_positions"] - g["body_positions"]).max() <= 3e-7
    assert np.mean(out["body_positions"] == g["body_positions"]) >= 0.999
    assert np.abs(out["body_rotations"] - g["body_rotations"]).max() <= 1e-7
    assert np.mean(out["body_rotations"] == g["body_rotations"]) >= 0.999
    assert np.all(np.sign(out["body_rotations"][..., 0]) == np.sign(g["body_rotations"][..., 0]))  # Eigen's sign convention
    assert np.abs(out["body_linear_velocities"] - g["body_linear_velocities"]).max() <= 2e-5  # 1-ulp positions / (2 dt)
    assert np.mean(out["body_linear_velocities"] == g["body_linear_velocities"]) >= 0.995
    err = np.abs(out["body_angular_velocities"].astype(np.float64) - g["body_angular_velocities"])
    assert np.mean(err <= 1e-5) >= 0.995       # measured 0.998
    assert err.max() <= max(ang_tol, 2.5e-2)   # arccos near w = 1: 0.02 rad/s is one ulp of the float32 quaternion
    # --- against the oracle (same arithmetic, host libm)
    assert np.abs(out["dof_positions"] - ref["dof_positions"]).max() <= 1e-15
    assert np.abs(out["body_positions"] - ref["body_positions"]).max() <= 3e-7
    # --- the written file loads through the engine's MotionLoader like the reference's
    path = str(tmp_path / "clip.npz")
    save_motion_npz(path, out)
    ml = MotionLoader(path, "cuda:0")
    assert ml.num_frames == out["dof_positions"].shape[0] and ml.dof_names == joints and ml.body_names == bodies


def test_converter_input_validation():
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.motions.convert import G1_CSV_JOINTS, G1_MODEL, MotionConverter

    conv = MotionConverter(G1_MODEL, G1_CSV_JOINTS, ["pelvis", "left_rubber_hand"], "cuda:0")
    with pytest.raises(nat.AmpEngineError):
        conv.convert(np.zeros((1, 36), dtype=np.float32))        # a single row cannot be up-sampled
    with pytest.raises(nat.AmpEngineError):
        conv.convert(np.zeros((5, 30), dtype=np.float32))        # wrong column count
    with pytest.raises(ValueError):
        MotionConverter(G1_MODEL, G1_CSV_JOINTS, ["no_such_link"], "cuda:0")
    rows = np.zeros((4, 36), dtype=np.float32)
    rows[:, 6] = 1.0                                             # identity root quaternion, zero pose
    out = conv.convert(rows)
    assert out["dof_positions"].shape == (7, 29) and np.all(out["dof_velocities"] == 0.0)
    assert np.allclose(out["body_rotations"][:, 0], [1, 0, 0, 0]) and np.all(out["body_angular_velocities"] == 0.0)
