"""Training step A/B on one box: python tools/timeline/train_ab.py [in_dim] [B]
gemm_precision f32 vs f16x3, deferred refresh, interleaved, best of three probes of 40 steps each + per-kernel brackets."""
import sys, time
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer
from humanoid_amp_amd.workloads import make_disc_weights

in_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 166
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
w = make_disc_weights(in_dim, 0)
g = torch.Generator().manual_seed(0)
pc, rc, mc = (torch.randn(B, in_dim, generator=g).cuda() for _ in range(3))
trainers = {}
for prec in ("f32", "f16x3"):
    disc = AmpDiscriminator(w, "cuda:0", running_mean=torch.zeros(in_dim, dtype=torch.float64), running_variance=torch.ones(in_dim, dtype=torch.float64))
    trainers[prec] = (disc, AmpDiscriminatorTrainer(disc, batch_size=B, defer_refresh=True, gemm_precision=prec))
for _, tr in trainers.values():
    for _ in range(10):
        tr.step(pc, rc, mc)
torch.cuda.synchronize()
best = {k: 1e9 for k in trainers}
for rep in range(3):
    for prec, (_, tr) in trainers.items():
        t0 = time.perf_counter()
        for _ in range(40):
            tr.step(pc, rc, mc)
        torch.cuda.synchronize()
        best[prec] = min(best[prec], (time.perf_counter() - t0) / 40 * 1e3)
for prec, (_, tr) in trainers.items():
    with nat.KernelTrace(4096) as trc:
        tr.step(pc, rc, mc)
        torch.cuda.synchronize()
    kern = {k: (c, round(t * 1e3, 1)) for k, (c, t) in trc.summary().items()}
    print(f"in_dim {in_dim} B {B} {prec}: {best[prec]:.3f} ms per step (deferred refresh)  kernels (calls, us): {kern}", flush=True)
