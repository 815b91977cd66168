"""Per-kernel times (HIP events around every launch) of one hot-path step, per discriminator engine / fusion mode."""
import contextlib, io, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.workloads import WORKLOADS, HotPath

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = sys.argv[2] if len(sys.argv) > 2 else "g1_walk"
for prec in ("f16x3", "f32"):
    for fused in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            hot = HotPath(WORKLOADS[wl], envs, "cuda:0", seed=1, disc_precision=prec, fused_scaler=fused)
        for _ in range(5):
            hot.step()
        torch.cuda.synchronize()
        with nat.KernelTrace(capacity=16 * 20) as tr:
            for _ in range(10):
                hot.step()
        s = {k: round(t / c * 1e3, 1) for k, (c, t) in tr.summary().items()}
        print(json.dumps({"engine": prec, "fused_scaler": fused, "sum_us": round(sum(s.values()), 1), **s}))
        del hot
        torch.cuda.empty_cache()
