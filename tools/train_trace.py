import sys, torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer
from humanoid_amp_amd.workloads import make_disc_weights
in_dim, B = 166, 4096
w = make_disc_weights(in_dim, 0)
disc = AmpDiscriminator(w, "cuda:0", running_mean=torch.zeros(in_dim, dtype=torch.float64), running_variance=torch.ones(in_dim, dtype=torch.float64))
tr = AmpDiscriminatorTrainer(disc, batch_size=B)
x = [torch.randn(B, in_dim, device="cuda") for _ in range(3)]
for _ in range(3): tr.step(*x)
with nat.KernelTrace(4096) as t:
    tr.step(*x)
names = ["fwd L1 [12288x176]x[1024]", "fwd L2 [12288x1024]x[512]", "dH1 = dH2 W2 (M=12288,N=1024,K=512)", "gW2 = dH2^T H1 (M=512,N=1024,K=12288)",
         "gW1 = dH1^T Xs (M=1024,N=192,K=12288)", "a1 = a2 W2 (M=4096,N=1024,K=512)", "g = a1 W1 (M=4096,N=192,K=1024)",
         "gW1 += a1^T dg (M=1024,N=192,K=4096)", "e1 = dg W1^T (M=4096,N=1024,K=176)", "gW2 += a2^T e1 (M=512,N=1024,K=4096)",
         "da2 = e1 W2^T (M=4096,N=512,K=1024)"]
i = 0
for name, ms in t.records():
    if "gemm" in name:
        print(f"{ms*1e3:8.1f} us  {name:24s} {names[i]}"); i += 1
