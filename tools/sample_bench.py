"""MotionLoader.sample (full 6-tuple surface) and the other BASELINE workloads -- run on the GPU box."""
import contextlib, json, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.motions import MotionLoader, MOTIONS_DIR

for clip, n in (("G1_walk", 131072), ("G1_dance", 81920), ("humanoid_walk,humanoid_run,humanoid_dance", 65536)):
    files = ",".join(f"{MOTIONS_DIR}/{c}.npz" for c in clip.split(","))
    with contextlib.redirect_stdout(sys.stderr):
        ml = MotionLoader(files, "cuda:0")
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, ml.num_trajectories, (n,), generator=g)
    t = (torch.rand(n, generator=g, dtype=torch.float64) * torch.from_numpy(ml.durations)[ids]).cuda()
    ids = ids.cuda()
    for _ in range(3):
        ml.sample(n, times=t, motion_ids=ids)
    with nat.KernelTrace(64) as tr:
        for _ in range(10):
            ml.sample(n, times=t, motion_ids=ids)
    c, ms = tr.summary()["sample_kernel"]
    us = ms / c * 1e3
    out_bytes = n * (2 * ml.num_dofs + 13 * ml.num_bodies) * 4
    print(json.dumps({"clip": clip, "n": n, "bodies": ml.num_bodies, "sample_kernel_us": round(us, 1),
                      "GB_written": round(out_bytes / 1e9, 3), "GBps": round(out_bytes / us / 1e3, 1)}))
