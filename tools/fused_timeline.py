"""In-kernel clock and k-loop time of the fused two-layer discriminator kernel: python tools/fused_timeline.py [rows]
(diagnostic build: tools/build_variant.sh fused_tl disc.hip -DAMP_FUSED_TIMELINE; run with AMP_ENGINE_LIB=tools/bin/libamp_fused_tl.so).
Thread 0 of every workgroup stores s_memtime (shader clock ticks) and s_memrealtime (100 MHz) in front of and behind the k-loop:
clock = d(memtime) / d(memrealtime) x 100 MHz, after >= 1 s of back-to-back launches on random data."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.engine import AmpDiscriminator
from humanoid_amp_amd.workloads import make_disc_weights

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
in_dim = 166
lib = nat.load()
d = AmpDiscriminator(make_disc_weights(in_dim, 0), "cuda:0", running_mean=torch.zeros(in_dim, dtype=torch.float64),
                     running_variance=torch.ones(in_dim, dtype=torch.float64))
x = torch.randn(rows, in_dim, device="cuda") * 1.5
t_end = time.time() + 1.5
while time.time() < t_end:   # let the clocks settle under the kernel's own load
    for _ in range(20):
        d.style_reward(x)
    torch.cuda.synchronize()
n_wg = (rows + 127) // 128
buf = torch.zeros(n_wg * 4, dtype=torch.int64, device="cuda")
lib.amp_debug_fused_timeline.argtypes = [C.c_void_p]
assert lib.amp_debug_fused_timeline(C.c_void_p(buf.data_ptr())) == 0
for _ in range(5):
    d.style_reward(x)
torch.cuda.synchronize()
assert lib.amp_debug_fused_timeline(C.c_void_p(0)) == 0
t = buf.view(n_wg, 4).cpu().numpy().astype(np.float64)
cyc, real = t[:, 2] - t[:, 0], (t[:, 3] - t[:, 1]) * 10.0   # ns
ok = (real > 0) & (cyc > 0)
mhz = cyc[ok] / real[ok] * 1e3
print(f"{ok.sum()} workgroups: k-loop {np.median(real[ok]) / 1e3:.1f} us (p10 {np.percentile(real[ok], 10) / 1e3:.1f}, p90 {np.percentile(real[ok], 90) / 1e3:.1f}); "
      f"shader clock {np.median(mhz):.0f} MHz (p10 {np.percentile(mhz, 10):.0f}, p90 {np.percentile(mhz, 90):.0f})")
start = (t[:, 1] - t[:, 1].min()) * 0.01
print("k-loop start times (us), deciles:", np.round(np.percentile(start, np.arange(0, 101, 10)), 1))
print(f"MFMA cycles per k-loop and SIMD: 32 k-blocks x 264 x 16 = {32 * 264 * 16}; pipe busy inside the loop = {32 * 264 * 16 / np.median(cyc[ok]):.3f}")
