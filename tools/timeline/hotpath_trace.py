"""A few hot-path steps (amp_hot_step) for a rocprofv3 --kernel-trace timeline: python tools/timeline/hotpath_trace.py [envs] [workload] [graph]"""
import sys
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd.workloads import WORKLOADS, HotPath

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
wl = sys.argv[2] if len(sys.argv) > 2 else "g1_walk"
graph = len(sys.argv) > 3 and sys.argv[3] == "1"
hot = HotPath(WORKLOADS[wl], envs, "cuda:0", seed=1, state_sets=8)
if graph:
    hot.capture()
for _ in range(40):
    hot.step()
torch.cuda.synchronize()
