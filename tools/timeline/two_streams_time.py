"""HotPath default (one stream) vs two_streams=True (disc half of step t beside the env launch of step t + 1): ms per step."""
import sys, time
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd.workloads import WORKLOADS, HotPath

for envs in (65536, 32768, 8192):
    for name, kw in (("one stream", {}), ("two streams", dict(two_streams=True)), ("one stream", {}), ("two streams", dict(two_streams=True))):
        hot = HotPath(WORKLOADS["g1_walk"], envs, "cuda:0", seed=1, state_sets=8, **kw)
        for _ in range(20):
            hot.step()
        hot.synchronize() if hasattr(hot, "synchronize") else None
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            hot.step()
        hot.synchronize() if hasattr(hot, "synchronize") else None
        torch.cuda.synchronize()
        print(envs, name, round((time.perf_counter() - t0) / n * 1e6, 2), "us/step", flush=True)
        del hot
