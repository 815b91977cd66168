"""Env-step launch alone (and fused with the expert sample), HIP-event timed: iterate on env_step.hip without the GEMMs.

    python tools/env_step_bench.py [envs] [workload]
"""
import contextlib, io, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.workloads import WORKLOADS, HotPath, algorithmic_bytes_per_env_step

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = sys.argv[2] if len(sys.argv) > 2 else "g1_walk"
spec = WORKLOADS[wl]
with contextlib.redirect_stdout(io.StringIO()):
    hot = HotPath(spec, envs, "cuda:0", seed=1)
k, s = hot.kernel, hot.state


flush = torch.empty(160 * 1024 * 1024, dtype=torch.float32, device="cuda:0")  # 640 MB: evicts L2 + Infinity Cache


def run(reference, phases=nat.AMP_PHASE_ALL, iters=30, cold=False):
    for _ in range(5):
        k.launch(phases, key_body_indexes=[0, 1, 2, 3], reference=reference, **hot._sim)
    torch.cuda.synchronize()
    with nat.KernelTrace(capacity=4 * iters) as tr:
        for _ in range(iters):
            if cold:  # as inside the step, where the GEMMs have streamed ~1 GB through the caches since the last launch
                flush.fill_(0.0)
            k.launch(phases, key_body_indexes=[0, 1, 2, 3], reference=reference, **hot._sim)
    return {name: round(t / c * 1e3, 2) for name, (c, t) in tr.summary().items()}


ref = (hot.motion, s["motion_times"], s["motion_ids"], hot.expert_obs)
out = {"envs": envs, "workload": wl, "alg_bytes_per_env": algorithmic_bytes_per_env_step(spec)}
out["env_only_us"] = run(None)
out["fused_us"] = run(ref)
out["env_only_cold_us"] = run(None, cold=True)
out["fused_cold_us"] = run(ref, cold=True)
out["obs_only_us"] = run(None, nat.AMP_PHASE_OBS)
out["dones_reward_us"] = run(None, nat.AMP_PHASE_DONES | nat.AMP_PHASE_REWARD)
print(json.dumps(out))
