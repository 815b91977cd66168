"""Race screen for the LDS-DMA GEMM kernels: many launches at many (ragged) shard sizes, each compared row by row with
the fp32-MFMA engine (a different kernel family with ordinary register staging).  A mis-ordered LDS-DMA read shows up
as a wrong tile -- errors of order one, not 1e-6 -- possibly only once in many launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanoid_amp_amd.engine import AmpDiscriminator
from humanoid_amp_amd.workloads import make_disc_weights

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
torch.manual_seed(0)
worst, launches, t_end = 0.0, 0, time.time() + seconds
for in_dim in (166, 830, 162, 142, 100):
    w = make_disc_weights(in_dim, 0)
    kw = dict(running_mean=torch.zeros(in_dim, dtype=torch.float64), running_variance=torch.ones(in_dim, dtype=torch.float64))
    fast = AmpDiscriminator(w, "cuda:0", precision="f16x3", **kw)
    slow = AmpDiscriminator(w, "cuda:0", precision="f32", **kw)
    g = torch.Generator(device="cuda").manual_seed(in_dim)
    deadline = time.time() + seconds / 5
    t_print = time.time()
    while time.time() < deadline:
        if time.time() - t_print > 30:  # a run that stays silent for minutes is taken to be hung by the GPU harness
            print(f"  ... {launches} passes, worst {worst:.3e}", flush=True)
            t_print = time.time()
        # half of the draws from the large-shard plans (256 x 256 tiles, 32 768-row chunks), half from the small ones
        # (64 x 128 / 128 x 128 / 256 x 128 tiles, the k-block-per-segment ring, persistent layer-1 workgroups, register-staged)
        rows = int(torch.randint(24576, 90000, (1,)).item()) if launches % 8 < 4 else int(torch.randint(1, 24576, (1,)).item())
        x = torch.randn(rows, in_dim, device="cuda", generator=g) * 1.5
        ref = slow.style_reward(x, want_logits=True)["logits"]
        for _ in range(4):  # the same input several times: a race would make the launches disagree
            got = fast.style_reward(x, want_logits=True)["logits"]
            err = float((got - ref).abs().max())
            worst = max(worst, err)
            launches += 1
            if err > 1e-4:
                print(f"MISMATCH rows={rows} in_dim={in_dim} err={err}")
                sys.exit(1)
print(f"race screen: {launches} forward passes at random shard sizes, worst |f16x3 - f32| = {worst:.3e}: clean")
