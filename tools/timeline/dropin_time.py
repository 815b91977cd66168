"""Drop-in env step, eager and captured: python tools/timeline/dropin_time.py"""
import sys, json
import torch
sys.path.insert(0, ".")
import bench
from humanoid_amp_amd.workloads import WORKLOADS
for wl, n in (("g1_walk", 8192), ("g1_walk", 4096), ("g1_walk", 65536), ("g1_dance", 8192)):
    e = bench.dropin_env_step(WORKLOADS[wl], n, "cuda:0")
    print(wl, n, "eager", round(e["eager"]["us_per_step"], 1), "graph", round(e["hipgraph"]["us_per_step"], 1), flush=True)
