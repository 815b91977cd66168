#!/usr/bin/env python3
"""Golden vectors for the CSV -> npz converter: DATA of the reference only.

Inputs: rows of datasets/walk1_subject1.csv.  Expected outputs: the arrays of the reference's own shipped clips, which
are motions/data_convert.py run on exactly those rows (found by matching joint angles):
    motions/G1_walk.npz        = rows [100:300], 11 bodies   (the BASELINE headline clip)
    motions/custom_motion.npz  = rows [110:265], 25 bodies
Nothing of the reference is executed.  Run in the survey container: python tests/golden/gen_convert_golden.py
"""
import os

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
csv = np.loadtxt(os.path.join(REF, "datasets", "walk1_subject1.csv"), delimiter=",", dtype=np.float32)
for name, clip, lo, hi in (("convert_g1_walk", "G1_walk.npz", 100, 300), ("convert_custom_motion", "custom_motion.npz", 110, 265)):
    ref = np.load(os.path.join(REF, "motions", clip))
    first = np.abs(csv[:, 7:] - ref["dof_positions"][0].astype(np.float32)).max(1).argmin()
    assert first == lo and ref["dof_positions"].shape[0] == 2 * (hi - lo) - 1, (first, ref["dof_positions"].shape)
    out = {"csv_rows": csv[lo:hi], "row_range": np.array([lo, hi])}
    out.update({k: ref[k] for k in ref.files})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})
