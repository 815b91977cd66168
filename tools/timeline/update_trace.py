"""One AMP discriminator update at the reference's sizes (16 rollouts x 4096 envs, batch 4096, 6 epochs x 2 minibatches; replay 1 M rows,
motion dataset 200 k rows) -- wall time per update and, under rocprofv3 --kernel-trace, its kernels: python tools/timeline/update_trace.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer, AmpDiscriminatorUpdate, AmpReplayBuffer
from humanoid_amp_amd.workloads import make_disc_weights

C, bs = 166, 4096
disc = AmpDiscriminator(make_disc_weights(C, 0), "cuda:0", running_mean=torch.zeros(C, dtype=torch.float64), running_variance=torch.ones(C, dtype=torch.float64))
trainer = AmpDiscriminatorTrainer(disc, batch_size=bs, defer_refresh=True)
replay, motion = AmpReplayBuffer(1_000_000, C, "cuda:0", seed=5), AmpReplayBuffer(200_000, C, "cuda:0", seed=6)
g = torch.Generator(device="cuda").manual_seed(0)
motion.add_samples(torch.randn(200_000, C, device="cuda", generator=g))
upd = AmpDiscriminatorUpdate(trainer, replay, motion, learning_epochs=6, mini_batches=2, seed=9)
rollout = torch.randn(16, 4096, C, device="cuda", generator=g)
for _ in range(2):
    upd.update(rollout)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    upd.update(rollout)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"update: {ms:.3f} ms for 12 training steps = {ms / 12:.4f} ms per step incl. draws, shuffles, replay append")
