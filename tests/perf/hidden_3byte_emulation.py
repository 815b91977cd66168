"""could the hidden layer be stored in 3 bytes per element (fp16 p0 + an 8-bit residual) and still meet
tests/test_gpu_disc.py::test_gemm_engines_vs_fp64's bar (|logit - fp64| <= 1e-6 * max(1, |logit|))?  numpy emulation of the
engine's arithmetic on the CPU: layer 1 exact (fp64), hidden activations rounded to the candidate storage format, layer 2 + the
512 -> 1 layer in fp64.  The only error is the storage format's, so this is a LOWER bound on what the kernel would show."""
import numpy as np
import torch
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import disc as odisc

rng = np.random.default_rng(0)
in_dim, M = 166, 4096
w = [(a.double().numpy(), b.double().numpy()) for a, b in odisc.make_weights(in_dim, seed=0)]
x = np.clip(rng.normal(0, 1.5, size=(M, in_dim)), -5, 5)
h1 = np.maximum(x @ w[0][0].T + w[0][1], 0.0)

def logits(h):
    h2 = np.maximum(h @ w[1][0].T + w[1][1], 0.0)
    return h2 @ w[2][0].T + w[2][1]

ref = logits(h1)
bound = np.abs(h1).max()
s = 2.0 ** np.floor(np.log2(32768.0 / bound))            # the engine's power-of-two plane scale
p0 = (h1 * s).astype(np.float16).astype(np.float64)
res = h1 * s - p0
def report(name, h):
    err = np.abs(logits(h) - ref)
    bar = 1e-6 * np.maximum(1.0, np.abs(ref))
    print(f"{name:44s} max |err| {err.max():.3e}   worst err / bar {np.max(err / bar):8.2f}   rows over the bar {int((err > bar).any(axis=1).sum())} / {M}")
report("fp16 p0 + fp16 p1 (today, 4 B / element)", (p0 + res.astype(np.float16).astype(np.float64)) / s)
# 8-bit residual, best case: per-element exponent taken from p0 (residual in units of ulp(p0) / 256), i.e. int8 of the 11 low bits' top 8
ulp = np.spacing(np.abs(p0).astype(np.float16)).astype(np.float64)
ulp[ulp == 0] = np.spacing(np.float16(0)).astype(np.float64)
r8 = np.clip(np.rint(res / (ulp / 256.0)), -128, 127) * (ulp / 256.0)
report("fp16 p0 + int8 residual in ulp(p0)/256 (3 B)", (p0 + r8) / s)
# fp8 e4m3 residual with a per-tensor power-of-two scale
def e4m3(v):
    t = torch.from_numpy(v).to(torch.float32)
    return t.to(torch.float8_e4m3fn).to(torch.float64).numpy()
sc = 2.0 ** np.floor(np.log2(448.0 / max(np.abs(res).max(), 1e-30)))
report("fp16 p0 + fp8 e4m3 residual (3 B)", (p0 + e4m3(res * sc) / sc) / s)
report("fp16 p0 only (2 B)", p0 / s)
