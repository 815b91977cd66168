"""Time amp_env_step per phase (HIP-event tracer) -- run on the GPU box."""
import contextlib, sys
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.workloads import WORKLOADS, HotPath

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = sys.argv[2] if len(sys.argv) > 2 else "g1_walk"
with contextlib.redirect_stdout(sys.stderr):
    hot = HotPath(WORKLOADS[wl], envs, "cuda:0", seed=1)
k = hot.kernel
for name, ph in (("dones", 1), ("reward", 2), ("obs", 4), ("dones+reward", 3), ("all", 7)):
    for _ in range(5):
        k.launch(ph, key_body_indexes=[0, 1, 2, 3], **hot._sim)
    with nat.KernelTrace(64) as tr:
        for _ in range(20):
            k.launch(ph, key_body_indexes=[0, 1, 2, 3], **hot._sim)
    c, t = tr.summary()["env_step_kernel"]
    print(f"{envs:6d} {wl} {name:13s} {t / c * 1e3:7.2f} us")
