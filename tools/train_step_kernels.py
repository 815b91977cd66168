"""Per-kernel HIP-event times of one discriminator training step (3 x 4096 rows): python tools/train_step_kernels.py [f32|f16x3]"""
import sys, torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer
from humanoid_amp_amd.workloads import make_disc_weights
in_dim, B = 166, 4096
w = make_disc_weights(in_dim, 0)
disc = AmpDiscriminator(w, "cuda:0", running_mean=torch.zeros(in_dim, dtype=torch.float64), running_variance=torch.ones(in_dim, dtype=torch.float64))
tr = AmpDiscriminatorTrainer(disc, batch_size=B, gemm_precision=sys.argv[1] if len(sys.argv) > 1 else "f32")
x = [torch.randn(B, in_dim, device="cuda") for _ in range(3)]
for _ in range(3): tr.step(*x)
torch.cuda.synchronize()
import time
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): tr.step(*x)
e1.record(); torch.cuda.synchronize()
print("untraced ms/step", e0.elapsed_time(e1) / 20)
with nat.KernelTrace(4096) as t:
    tr.step(*x)
tot = 0
for name, ms in t.records():
    print(f"{ms*1e3:8.1f} us  {name}")
    tot += ms
print("kernel sum ms", tot, "launches", len(t.records()))
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): tr.step(*x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue ms/step", (t1 - t0) / 50 * 1e3, " wall ms/step incl. drain", (t2 - t0) / 50 * 1e3)
