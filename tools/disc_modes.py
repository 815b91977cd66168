"""Per-kernel times of the discriminator in each precision mode (run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.engine import AmpDiscriminator
from humanoid_amp_amd.workloads import make_disc_weights

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
w = make_disc_weights(166, 0)
x = torch.randn(rows, 166, device="cuda")
for mode in ("f16x3", "f32"):
    d = AmpDiscriminator(w, "cuda:0", running_mean=torch.zeros(166, dtype=torch.float64),
                         running_variance=torch.ones(166, dtype=torch.float64), precision=mode)
    for _ in range(3):
        d.style_reward(x)
    with nat.KernelTrace(256) as tr:
        for _ in range(10):
            d.style_reward(x)
    print(mode, {k: round(t / c * 1e3, 1) for k, (c, t) in tr.summary().items()})
