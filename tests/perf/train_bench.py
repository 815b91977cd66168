"""Discriminator training step: GPU time per kernel vs the torch-autograd oracle on the host (run on the GPU box)."""
import json, sys, time
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer
from humanoid_amp_amd.workloads import make_disc_weights
from oracle import disc_train as odt

in_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 166
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"   # the trainer's default
w = make_disc_weights(in_dim, 0)
disc = AmpDiscriminator(w, "cuda:0", running_mean=torch.zeros(in_dim, dtype=torch.float64), running_variance=torch.ones(in_dim, dtype=torch.float64))
tr = AmpDiscriminatorTrainer(disc, batch_size=B, gemm_precision=prec)
g = torch.Generator().manual_seed(0)
p, r, m = (torch.randn(B, in_dim, generator=g) for _ in range(3))
pc, rc, mc = p.cuda(), r.cuda(), m.cuda()
for _ in range(10):   # (three warm-up steps left the first timed mode ~50 us per step slower than the later ones: clocks, allocator)
    tr.step(pc, rc, mc)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 40
for _ in range(n):
    tr.step(pc, rc, mc)
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / n * 1e3
tr.capture()
for _ in range(3):
    tr.step_captured(pc, rc, mc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    tr.step_captured(pc, rc, mc)
torch.cuda.synchronize()
graph_ms = (time.perf_counter() - t0) / n * 1e3
disc2 = AmpDiscriminator(w, "cuda:0", running_mean=torch.zeros(in_dim, dtype=torch.float64), running_variance=torch.ones(in_dim, dtype=torch.float64))
tr2 = AmpDiscriminatorTrainer(disc2, batch_size=B, defer_refresh=True, gemm_precision=prec)
for _ in range(10):
    tr2.step(pc, rc, mc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    tr2.step(pc, rc, mc)
tr2.refresh()
torch.cuda.synchronize()
deferred_ms = (time.perf_counter() - t0) / n * 1e3
with nat.KernelTrace(4096) as trc:
    tr.step(pc, rc, mc)
kern = {k: (c, round(t * 1e3, 1)) for k, (c, t) in trc.summary().items()}
# algorithmic FLOPs: forward 3B rows, backward 2x forward, gradient penalty 6 GEMMs on B rows
fwd = 2.0 * 3 * B * (in_dim * 1024 + 1024 * 512 + 512)
gp = 2.0 * B * (2 * 512 * 1024 + 2 * 1024 * in_dim + 512 * 1024 + 1024 * in_dim)
flops = 3 * fwd + gp
torch.set_num_threads(16)
mean, var = torch.zeros(in_dim, dtype=torch.float64), torch.ones(in_dim, dtype=torch.float64)
odt.loss_and_grads(w, p, r, m, mean, var)
t0 = time.perf_counter()
for _ in range(3):
    odt.loss_and_grads(w, p, r, m, mean, var)
cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
print(json.dumps({"gemm_precision": prec, "in_dim": in_dim, "rows_per_group": B, "gpu_ms_per_step": round(gpu_ms, 3), "hipgraph_ms_per_step": round(graph_ms, 3), "deferred_refresh_ms_per_step": round(deferred_ms, 3), "tflops": round(flops / gpu_ms / 1e9, 1),
                  "cpu_autograd_ms_per_step_16thr": round(cpu_ms, 1), "speedup": round(cpu_ms / gpu_ms, 1), "kernels_calls_us": kern}))
