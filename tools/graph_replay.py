"""Replay the captured hot-path step of a small shard (for rocprofv3 --kernel-trace: durations and gaps between the nodes).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -- python3 tools/graph_replay.py 8192 300
"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanoid_amp_amd.workloads import WORKLOADS, HotPath

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
mode = sys.argv[3] if len(sys.argv) > 3 else "graph"   # graph | eager
graph = mode == "graph"
with contextlib.redirect_stdout(io.StringIO()):
    hot = HotPath(WORKLOADS[sys.argv[4] if len(sys.argv) > 4 else "g1_walk"], envs, "cuda:0", seed=1, state_sets=3)
if graph:
    hot.capture()
for _ in range(50):
    hot.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    hot.step()
torch.cuda.synchronize()
print(f"{envs} envs, {mode}: {(time.perf_counter() - t0) / steps * 1e6:.1f} us / step")
