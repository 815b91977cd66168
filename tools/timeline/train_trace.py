"""A few training steps for a rocprofv3 --kernel-trace timeline: python tools/timeline/train_trace.py [in_dim] [defer] [f32|f16x3]"""
import sys
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer
from humanoid_amp_amd.workloads import make_disc_weights

in_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 166
defer = len(sys.argv) > 2 and sys.argv[2] == "1"
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
B = 4096
disc = AmpDiscriminator(make_disc_weights(in_dim, 0), "cuda:0", running_mean=torch.zeros(in_dim, dtype=torch.float64),
                        running_variance=torch.ones(in_dim, dtype=torch.float64))
tr = AmpDiscriminatorTrainer(disc, batch_size=B, defer_refresh=defer, gemm_precision=prec)
g = torch.Generator().manual_seed(0)
p, r, m = (torch.randn(B, in_dim, generator=g).cuda() for _ in range(3))
for _ in range(12):
    tr.step(p, r, m)
torch.cuda.synchronize()
