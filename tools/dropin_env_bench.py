"""The drop-in env classes stepped the way skrl drives them (``env.step(actions)`` -> the DirectRLEnv hooks), timed.

For each env count: wall time per ``env.step()`` (device_reset on, no host sync inside the loop), the engine's own kernels
per step (HIP-event tracer: what the hooks launch), and next to it ``HotPath``'s env launch at the same size -- the
benchmark composition's env-side kernel the drop-in path is compared with.  Usage: dropin_env_bench.py [envs ...]
"""
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from humanoid_amp_amd import _native as nat


def dropin_env_step(num_envs: int, steps: int = 50, warmup: int = 10, physics: bool = True, task: str = "walk",
                    device: str = "cuda:0", graph: bool = False):
    from humanoid_amp_amd.envs import G1AmpEnv, G1AmpDanceEnvCfg, G1AmpDeployEnvCfg, G1AmpWalkEnvCfg

    cfg = {"walk": G1AmpWalkEnvCfg, "dance": G1AmpDanceEnvCfg, "deploy": G1AmpDeployEnvCfg}[task]()
    cfg.scene.num_envs = int(num_envs)
    cfg.sim.device = device
    with contextlib.redirect_stdout(io.StringIO()):
        env = G1AmpEnv(cfg, device_reset=True, reset_seed=0)
    if not physics:
        env.robot.step = lambda: None
    env.reset()
    # episode phases spread out as in a long run (otherwise all envs time out on the same step)
    env.episode_length_buf.copy_(torch.randint(0, env.max_episode_length, (num_envs,), device=device))
    acts = [torch.randn(num_envs, cfg.action_space, device=device) * 0.3 for _ in range(4)]
    for i in range(warmup):
        env.step(acts[i & 3])
    if graph:
        env.capture_step()
        for i in range(3):
            env.step(acts[i & 3])
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")  # a host sync inside step() would raise
    try:
        t0 = time.perf_counter()
        marks = [t0]
        for i in range(steps):
            env.step(acts[i & 3])
            marks.append(time.perf_counter())
    finally:
        torch.cuda.set_sync_debug_mode("default")
    submit = (time.perf_counter() - t0) / steps   # host time to enqueue a step
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    env._graph = None  # the tracer brackets eager launches
    with nat.KernelTrace(capacity=64 * 20) as tr:
        for i in range(10):
            env.step(acts[i & 3])
    per = {k: round(t / 10 * 1e3, 2) for k, (c, t) in tr.summary().items()}   # us per step (all launches of that kernel)
    calls = {k: c / 10 for k, (c, t) in tr.summary().items()}
    resets = float(env._kernel.reset_count.item())
    return {"envs": num_envs, "task": task, "physics": physics, "graph": graph, "wall_us_per_step": round(wall * 1e6, 1), "host_submit_us_per_step": round(submit * 1e6, 1),
            "host_submit_us_median_max": [round(sorted(b - a for a, b in zip(marks, marks[1:]))[len(marks) // 2] * 1e6, 1),
                                          round(max(b - a for a, b in zip(marks, marks[1:])) * 1e6, 1)],
            "env_steps_per_s": round(num_envs / wall, 1), "engine_kernel_us_per_step": round(sum(per.values()), 1),
            "engine_launches_per_step": sum(calls.values()), "kernels_us": per, "resets_last_step": resets}


def hotpath_env_launch(num_envs: int, workload: str = "g1_walk", device: str = "cuda:0"):
    from humanoid_amp_amd.workloads import WORKLOADS, HotPath

    with contextlib.redirect_stdout(io.StringIO()):
        hot = HotPath(WORKLOADS[workload], num_envs, device, seed=1, state_sets=3)
    for _ in range(6):
        hot.step()
    torch.cuda.synchronize()
    with nat.KernelTrace(capacity=16 * 20) as tr:
        for _ in range(12):
            hot.step()
    s = {k: round(t / c * 1e3, 2) for k, (c, t) in tr.summary().items()}
    return {k: v for k, v in s.items() if "env_step" in k or "tail" in k}


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 8192, 65536]
    task = next((a for a in sys.argv[1:] if a in ("walk", "dance", "deploy")), "walk")
    for n in sizes:
        for physics, graph in ((True, False), (False, False), (True, True), (False, True)):
            try:
                print(json.dumps(dropin_env_step(n, physics=physics, graph=graph, task=task)), flush=True)
            except Exception as e:  # e.g. a torch build that cannot capture a custom generator
                print(json.dumps({"envs": n, "physics": physics, "graph": graph, "error": repr(e)[:300]}), flush=True)
        print(json.dumps({"envs": n, "hotpath_kernels_us": hotpath_env_launch(n)}), flush=True)
        torch.cuda.empty_cache()
