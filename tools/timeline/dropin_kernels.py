"""Per-kernel brackets of the drop-in env step (eager) + us per captured step: python tools/timeline/dropin_kernels.py [envs]"""
import sys
import torch
sys.path.insert(0, ".")
import bench
from humanoid_amp_amd.workloads import WORKLOADS
for n in (8192, 65536):
    e = bench.dropin_env_step(WORKLOADS["g1_walk"], n, "cuda:0")
    print(n, "eager", round(e["eager"]["us_per_step"], 1), "graph", round(e["hipgraph"]["us_per_step"], 1), e.get("engine_kernels_us_per_step") or {k: v for k, v in e.items() if "kernel" in k}, flush=True)
