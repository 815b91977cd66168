"""Where does the host time of one env-step go?  (run on the GPU box)"""
import contextlib, sys, time
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.workloads import WORKLOADS, HotPath

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with contextlib.redirect_stdout(sys.stderr):
    hot = HotPath(WORKLOADS["g1_walk"], envs, "cuda:0", seed=1)
for _ in range(20):
    hot.step()
torch.cuda.synchronize()
s, k = hot.state, hot.kernel
parts = {
    "collect": lambda: hot.motion.collect_reference(s["motion_times"], s["motion_ids"], hot.spec.K, out=hot.expert_obs),
    "env_step": lambda: k.launch(nat.AMP_PHASE_ALL, key_body_indexes=[0, 1, 2, 3], **hot._sim),
    "compact": lambda: k.compact_resets(),
    "disc": lambda: hot.disc.style_reward(k.amp_observation_buffer.view(envs, -1), k.reward),
}
for name, fn in parts.items():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    host = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 200
    print(f"{name:10s} host-issue {host*1e6:8.1f} us/call   incl. GPU drain {tot*1e6:8.1f} us/call")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    hot.step()
host = (time.perf_counter() - t0) / 200
torch.cuda.synchronize()
print(f"step       host-issue {host*1e6:8.1f} us   wall {(time.perf_counter()-t0)/200*1e6:8.1f} us")
