import sys, time, torch
sys.path.insert(0, ".")
from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer
from humanoid_amp_amd.workloads import make_disc_weights
from humanoid_amp_amd import _native as nat
for prec in ("f32", "f16x3"):
    disc = AmpDiscriminator(make_disc_weights(166, 0), "cuda:0", running_mean=torch.zeros(166, dtype=torch.float64), running_variance=torch.ones(166, dtype=torch.float64))
    tr = AmpDiscriminatorTrainer(disc, batch_size=4096, defer_refresh=True, gemm_precision=prec)
    g = torch.Generator().manual_seed(0)
    p, r, m = (torch.randn(4096, 166, generator=g).cuda() for _ in range(3))
    for _ in range(5): tr.step(p, r, m)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tr.step(p, r, m)
    torch.cuda.synchronize()
    print(prec, round((time.perf_counter() - t0) / 30 * 1e3, 3), "ms")
    with nat.KernelTrace(4096) as trc:
        tr.step(p, r, m)
    print({k: (c, round(t * 1e3, 1)) for k, (c, t) in trc.summary().items()})
