"""The step's tail as two launches (finalize, reset-id compaction) instead of the fused one: which half costs what."""
import contextlib, io, sys, torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.workloads import WORKLOADS, HotPath
for wl in (sys.argv[1:] or ["g1_walk", "g1_dance"]):
    with contextlib.redirect_stdout(io.StringIO()):
        hot = HotPath(WORKLOADS[wl], 65536, "cuda:0", seed=1, fused_tail=False, one_call=False)
    for _ in range(5): hot.step()
    torch.cuda.synchronize()
    with nat.KernelTrace(512) as t:
        for _ in range(10): hot.step()
    print(wl, {k: round(v[1] / v[0] * 1e3, 2) for k, v in t.summary().items()}, "resets", int(hot.kernel.reset_count))
    del hot
