"""g1_dance (K = 10): the expert sample fused into the env launch vs its own launch vs its own launch on a side stream."""
import sys, time
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.workloads import WORKLOADS, HotPath

wl = sys.argv[1] if len(sys.argv) > 1 else "g1_dance"
for envs in (65536, 8192):
    for name, kw in (("fused", {}), ("separate", dict(fused_expert=False)), ("side stream", dict(fused_expert=False, expert_stream=True)), ("fused", {})):
        hot = HotPath(WORKLOADS[wl], envs, "cuda:0", seed=1, state_sets=4, **kw)
        for _ in range(10):
            hot.step()
        torch.cuda.synchronize()
        n = 40
        t0 = time.perf_counter()
        for _ in range(n):
            hot.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        with nat.KernelTrace(64) as tr:
            hot.step()
            torch.cuda.synchronize()
        k = {a: round(b * 1e3, 1) for a, (c, b) in tr.summary().items()}
        print(wl, envs, name, round(ms, 4), k, flush=True)
        del hot
