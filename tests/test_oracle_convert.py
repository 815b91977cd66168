"""CPU: the converter oracle (oracle/convert.py) against the reference's OWN output files.

The reference ships the raw dataset (datasets/walk1_subject1.csv) and clips that are motions/data_convert.py run on
row ranges of it: G1_walk.npz = rows [100:300] (the BASELINE headline clip), custom_motion.npz = rows [110:265]
(tests/golden/gen_convert_golden.py).  Forward kinematics here is our own (Pinocchio absent) over the JSON tree
extracted from the reference's URDF -- so these files pin FK, up-sampling, differencing, smoothing and the quaternion
conventions at once."""

import os

import numpy as np
import pytest

from oracle import convert as oc

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL = os.path.join(os.path.dirname(HERE), "humanoid_amp_amd", "motions", "models", "g1_29dof.json")


@pytest.mark.parametrize("fixture,promotion,ang_tol", [("convert_g1_walk", "numpy2", 2.5e-2), ("convert_custom_motion", "numpy1", 2.5e-2)])
def test_oracle_reproduces_the_shipped_clip(fixture, promotion, ang_tol):
    g = np.load(os.path.join(HERE, "golden", fixture + ".npz"))
    out = oc.convert(g["csv_rows"], oc.load_model(MODEL), [str(n) for n in g["dof_names"]], [str(n) for n in g["body_names"]],
                     fps=int(g["fps"]), promotion=promotion)
    assert out["dof_positions"].dtype == g["dof_positions"].dtype == np.float64
    assert np.abs(out["dof_positions"] - g["dof_positions"]).max() <= 1e-15
    assert np.abs(out["dof_velocities"] - g["dof_velocities"]).max() <= 1e-13
    assert np.array_equal(out["body_rotations"], g["body_rotations"])          # every quaternion, every sign
    assert np.abs(out["body_positions"] - g["body_positions"]).max() <= 4e-9   # <= half an ulp of float32 at 3 m
    assert np.mean(out["body_positions"] == g["body_positions"]) >= 0.999
    assert np.abs(out["body_linear_velocities"] - g["body_linear_velocities"]).max() <= 2e-7
    err = np.abs(out["body_angular_velocities"].astype(np.float64) - g["body_angular_velocities"])
    # near-identity rotations: one ulp of the float32 relative quaternion (np.linalg.norm = the host BLAS's float32 dot,
    # whose summation order is not portable) moves arccos by up to 3.5e-4 rad = 0.02 rad/s; on the survey host the
    # oracle reproduces G1_walk.npz to 5e-7 and custom_motion.npz bit for bit on 92 % of the entries
    assert err.max() <= ang_tol
    assert np.mean(err <= 1e-5) >= 0.997
