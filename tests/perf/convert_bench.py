"""CSV -> npz converter: achieved parity against the reference's shipped clips and throughput (run on the GPU box)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from humanoid_amp_amd.motions.convert import G1_MODEL, MotionConverter
from oracle import convert as oc

HERE = os.path.dirname(os.path.abspath(__file__))
for fixture, gen in (("convert_g1_walk", 2), ("convert_custom_motion", 1)):
    g = np.load(os.path.join(HERE, "..", "tests", "golden", fixture + ".npz"))
    joints, bodies = [str(n) for n in g["dof_names"]], [str(n) for n in g["body_names"]]
    conv = MotionConverter(G1_MODEL, joints, bodies, "cuda:0")
    out = conv.convert(g["csv_rows"], numpy_generation=gen)
    rep = {"clip": fixture, "frames": int(out["dof_positions"].shape[0]), "bodies": len(bodies)}
    for k in ("dof_positions", "dof_velocities", "body_positions", "body_rotations", "body_linear_velocities", "body_angular_velocities"):
        e = np.abs(out[k].astype(np.float64) - g[k].astype(np.float64))
        rep[k] = {"max_abs_err": float(e.max()), "bit_exact": round(float(np.mean(out[k] == g[k])), 4)}
    e = np.abs(out["body_angular_velocities"].astype(np.float64) - g["body_angular_velocities"])
    rep["body_angular_velocities"]["within_1e-5"] = round(float(np.mean(e <= 1e-5)), 4)
    print(json.dumps(rep))
# throughput: the whole dataset's length (7840 rows -> 15 679 frames), 25 bodies
g = np.load(os.path.join(HERE, "..", "tests", "golden", "convert_custom_motion.npz"))
rows = np.tile(g["csv_rows"], (51, 1))[:7840]
joints, bodies = [str(n) for n in g["dof_names"]], [str(n) for n in g["body_names"]]
conv = MotionConverter(G1_MODEL, joints, bodies, "cuda:0")
conv.convert(rows)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    conv.convert(rows)
gpu = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
oc.convert(rows[:400], oc.load_model(G1_MODEL), joints, bodies)
cpu = (time.perf_counter() - t0) * (7840 / 400)
print(json.dumps({"rows": 7840, "frames": 15679, "gpu_s_incl_h2d_d2h": round(gpu, 4), "cpu_oracle_s_extrapolated": round(cpu, 2),
                  "speedup": round(cpu / gpu, 1)}))
