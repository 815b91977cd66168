"""Per-workgroup phase stamps of the two discriminator GEMM launches of a small shard, inside the hot step:
python tools/small_shard_timeline.py [envs] [workload]
(diagnostic build: tools/build_variant.sh tl disc.hip -DAMP_DMA_TIMELINE; run with AMP_ENGINE_LIB=tools/bin/libamp_tl.so).
Thread 0 of every workgroup stores s_memrealtime (100 MHz) at its phase boundaries; times are us relative to the first layer-1
workgroup's start, as min / median / p90 / max over the workgroups (profiles/r05_small_shard.md)."""
import contextlib, ctypes as C, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.workloads import WORKLOADS, HotPath

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = sys.argv[2] if len(sys.argv) > 2 else "g1_walk"
lib = nat.load()
lib.amp_debug_dma_timeline.argtypes = [C.c_void_p]
with contextlib.redirect_stdout(io.StringIO()):
    hot = HotPath(WORKLOADS[wl], envs, "cuda:0", seed=1)


def stats(name, v):
    v = v[np.isfinite(v)]
    if v.size == 0:
        print(f"  {name:40s} (no stamps)")
        return
    print(f"  {name:40s} min {v.min():6.2f}  p50 {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")


for _ in range(50):
    hot.step()
torch.cuda.synchronize()
dma = torch.zeros(2048 * 8, dtype=torch.int64, device="cuda")
assert lib.amp_debug_dma_timeline(C.c_void_p(dma.data_ptr())) == 0
hot.step()
torch.cuda.synchronize()
assert lib.amp_debug_dma_timeline(C.c_void_p(0)) == 0
d = dma.view(-1, 8).cpu().numpy().astype(np.float64) * 0.01
d[d == 0] = np.nan
l1, l2 = d[:1024], d[1024:]
t0 = np.nanmin(l1[:, 0])
print(f"== {wl} {envs} envs: {hot.disc.plan_info(envs)['plan_name']}")
stats("layer 1: workgroup start", l1[:, 0] - t0); stats("layer 1: first k-block landed", l1[:, 1] - t0); stats("layer 1: k-loop end", l1[:, 2] - t0)
stats("layer 1: hidden-layer stores issued", l1[:, 3] - t0); stats("layer 1: stores acknowledged", l1[:, 4] - t0)
stats("layer 2: workgroup start", l2[:, 0] - t0); stats("layer 2: first k-block landed", l2[:, 1] - t0); stats("layer 2: k-loop end", l2[:, 2] - t0)
stats("layer 2: partial logits reduced", l2[:, 3] - t0)
stats("layer 1 k-loop (us)", l1[:, 2] - l1[:, 1]); stats("layer 2 k-loop (us)", l2[:, 2] - l2[:, 1])
stats("layer 1 workgroup life (us)", l1[:, 4] - l1[:, 0]); stats("layer 2 workgroup life (us)", l2[:, 3] - l2[:, 0])
