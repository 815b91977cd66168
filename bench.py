#!/usr/bin/env python3
"""bench.py -- AMP obs + motion-sample + reward env-steps/s on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one env-step of the hot path for every env of the shard (humanoid_amp_amd/workloads.py), inputs
resident in HBM.  One process per GPU, envs shard trivially, no data-path collective in the step.  The path's one exchange
belongs to the agent's discriminator update (agents/*.yaml:64-66,91): every rank contributes batch / world rows
of each of the update's 12 training steps, ONE RCCL all-gather per update, then every replica takes the same optimizer steps.
With --gpus > 1 one update runs every --rollouts steps INSIDE the timed region (its ms are reported separately in
config.update, with the value excluding it); on one GPU the timed region is the hot path alone (the metric's unit of work,
SURVEY 8d) and the update is timed right after it (--update-every R puts it inside there too).

Scaling mode.  BASELINE.json names GLOBAL env counts: configs[4] = G1-AMP-Walk, 65 536 envs on 1 and on 8 GPUs
(8 192 envs per GPU), configs[3] = humanoid 3-clip, 32 768 envs on 4 GPUs.  So the default is STRONG scaling of the
named configuration: `--global-envs` (default 65 536 for g1_walk / g1_dance, 32 768 for humanoid3) is split over
--gpus ranks by `distributed.shard_bounds`, and the line says "scaling": "strong" and `envs_per_gpu`.
    configs[4]:  torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8
    configs[3]:  torch.distributed.run --nproc-per-node 4 ... bench.py --gpus 4 --workload humanoid3
`--envs E` fixes the envs PER GPU instead (weak scaling, "scaling": "weak").
Rank 0 prints ONE JSON line (< 4 KB: `config`, `roofline` -- with the strict-fp32 engine and the 4 096- / 8 192-env shards
inside it -- and `cpu_baseline`); per-kernel tables, the drop-in env measurements and the other configs' kernel tables go to
stderr ("[bench detail] {...}") and gpurun_out/bench_detail.json.  The CPU oracle (oracle/) is used only for the bounded
`cpu_baseline` leg.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)
# layer 2 of the discriminator: 2*M*1024*512 of the 2*M*(in*1024+1024*512+512) FLOPs
# tracer labels: the fp16 engine launches disc_gemm_f16_dma_kernel<1> (LDS-DMA tiles) or disc_gemm_f16_kernel<1>
# (round 4: shards of >= 24 576 rows run both layers as ONE launch, disc_mlp_fused_kernel; the tracer filter matches either)
DOMINANT_FILTER = {"f16x3": "disc_", "f32": "disc_gemm_kernel<1>"}
FUSED_KERNEL = "disc_mlp_fused_kernel"


def is_dominant(name):
    """Tracer labels of the dominant kernel: the fused two-layer kernel, or a layer-2 GEMM launch."""
    return name == FUSED_KERNEL or (name.endswith("<1>") and "disc_gemm" in name)
DOMINANT_KERNEL = {"f16x3": "disc_gemm_f16_dma_kernel<1>", "f32": "disc_gemm_kernel<1>"}
MFMA_PEAK_TFLOPS = {"f16x3": 16 * 157.3, "f32": 157.3}  # dense fp16 MFMA = 16 x the fp32 MFMA rate (MI355X_MICROARCH.md)
MFMA_PER_PRODUCT = {"f16x3": 3, "f32": 1}               # the fp16 engine issues three MFMA products per algorithmic one
DTYPE = {"f16x3": "f32 (GEMM operands as 2 fp16 planes, 3 fp16 MFMAs per product, f32 accumulate)", "f32": "f32"}
DEFAULT_GLOBAL_ENVS = {"g1_walk": 65536, "g1_dance": 65536, "humanoid3": 32768}  # BASELINE.json configs[4] / [3]
BASELINE_CONFIG = {("g1_walk", 65536, 1): "configs[4] on 1 GPU", ("g1_walk", 65536, 8): "configs[4]",
                   ("humanoid3", 32768, 4): "configs[3]", ("g1_walk", 4096, 1): "configs[1]", ("g1_dance", 8192, 1): "configs[2]"}
# hipGraph replay is OPT-IN (--graph): with the whole step issued by one C call (amp_hot_step) the eager launches queue up
# back to back (rocprofv3: gaps <= 0.6 us), while a graph replay pays ~8.5 us between replays (profiles/r02_small_shard_gaps.md):
# 8 192 envs 63.6 us eager vs 67.7 us replayed, 4 096 envs 48.7 vs 53.2
GRAPH_MAX_ENVS = 16384       # --graph replays shards up to this size as one hipGraph
TRACE_EVERY = 5              # large shards: every 5th discriminator-GEMM launch of the timed region carries an event pair


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--global-envs", type=int, default=0,
                    help="envs of the whole job, split over --gpus ranks (strong scaling; 0 = the BASELINE config of the workload)")
    ap.add_argument("--envs", type=int, default=0, help="envs PER GPU (weak scaling); overrides --global-envs")
    ap.add_argument("--workload", default="g1_walk", choices=["g1_walk", "g1_dance", "humanoid3", "g1_walk_23dof"])
    ap.add_argument("--rollouts", type=int, default=16, help="env steps per rollout (agents/*.yaml:64)")
    ap.add_argument("--update-every", type=int, default=-1,
                    help="run one discriminator update every this many steps INSIDE the timed region (0: never; -1 = auto: --rollouts when "
                         "--gpus > 1, 0 on one GPU where the update is timed right after the region instead)")
    ap.add_argument("--batch", type=int, default=4096, help="GLOBAL discriminator minibatch (discriminator_batch_size, agents/*.yaml:91)")
    ap.add_argument("--epochs", type=int, default=6, help="learning_epochs (agents/*.yaml:65)")
    ap.add_argument("--mini-batches", type=int, default=2, help="mini_batches (agents/*.yaml:66)")
    ap.add_argument("--no-update", action="store_true", help="skip the agent side (rollout store + discriminator update) altogether")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=0, help="envs of the CPU sample (0 = same as the shard)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 8192- / 4096-env secondary measurements")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in env (G1AmpEnv.step through the hooks) measurements")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the bounded entries for BASELINE.json configs[2] / configs[3] (g1_dance K = 10, humanoid 3-clip)")
    ap.add_argument("--disc-precision", default="f16x3", choices=["f16x3", "f32"],
                    help="GEMM engine of the discriminator (both fp32-class accuracy): fp16-split (default) or fp32 MFMA")
    ap.add_argument("--no-fp32-engine", action="store_true", help="skip the comparison run on the fp32-MFMA GEMM engine")
    ap.add_argument("--graph", action="store_true", help="replay shards of <= 16384 envs as a captured hipGraph instead of eager launches")
    ap.add_argument("--state-sets", type=int, default=0,
                    help="synthetic input sets visited round-robin (0 = enough to exceed the 256 MB Infinity Cache, >= 3)")
    return ap.parse_args()


def n_state_sets(spec, envs, requested):
    """>= 3 input sets and at least 320 MB of rotating state, so that a step's state reads come from HBM rather than from
    the Infinity Cache lines the same addresses left there one step earlier (capped at 8 sets)."""
    if requested > 0:
        return requested
    per_env = (4 * spec.n_dof + 3 + 4 + 3 + 3 + 12 + 2) * 4 + 8 + 16  # the SoA rows a step reads
    return int(min(8, max(3, -(-320e6 // max(per_env * envs, 1)))))


def settle(hot, max_seconds=4.0, block=20):
    """Untimed pre-warm-up: a fresh box (first process after the image is paged in, clocks ramping, other tenants on the
    host) can run the first seconds at a fraction of the steady rate.  Blocks of `block` steps are run until two
    consecutive blocks are within 5 % of the fastest block seen (bounded by max_seconds); no collectives, so ranks may
    leave at different block counts.  The contract's W warm-up steps and K timed steps follow unchanged."""
    best, t_end, stable = float("inf"), time.perf_counter() + max_seconds, 0
    while time.perf_counter() < t_end:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(block):
            hot.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = min(best, dt)
        stable = stable + 1 if dt <= 1.05 * best else 0
        if stable >= 2:
            break


def probe_ms(hot, steps=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        hot.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def choose_launch_mode(hot, envs, force_graph):
    """Small shards (<= GRAPH_MAX_ENVS envs): eager one-call launches are normally ~4 us per step faster than replaying a
    captured hipGraph (no 8.5-us gap between replays), but they depend on the host keeping four launches per ~60 us ahead
    of the GPU -- on a box whose host is busy the eager step has been seen at 1.7 ms.  So both are probed (30 steps each,
    untimed) and the faster one is kept; the choice and both probes are reported."""
    if envs > GRAPH_MAX_ENVS:
        return "eager: one amp_hot_step call per step (4 kernel launches)", None
    for _ in range(10):
        hot.step()
    eager = min(probe_ms(hot), probe_ms(hot))
    hot.capture()
    graph = min(probe_ms(hot), probe_ms(hot))
    probes = {"eager_ms_per_step": eager, "graph_ms_per_step": graph}
    if force_graph or graph < eager:
        return "hipGraph replay of the captured step", probes
    hot._graphs = None
    return "eager: one amp_hot_step call per step (4 kernel launches)", probes


def timed_steps(hot, steps, warmup, world, after_step=None):
    """W untimed warm-up steps, then exactly K timed steps between barrier + synchronize brackets; max over ranks.  `after_step(i)`
    (the agent side: rollout store + the discriminator update every --update-every steps) runs inside both loops, so whatever it
    enqueues -- the update's collective included -- completes inside the timed region."""
    import torch.distributed as dist

    for i in range(warmup):
        hot.step()
        if after_step:
            after_step(i - warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        hot.step()
        if after_step:
            after_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(spec, n_envs, seed, target_seconds=12.0, probe=True, threads=None, min_reps=3):
    """The oracle (CPU restatement of the reference's torch path) timed on this box's host cores: same unit of
    work, same synthetic inputs; bounded to ~target_seconds."""
    # a one-GPU box owns a 16-core share of the host (more threads only oversubscribe it)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, 16) if threads is None else int(threads)
    torch.set_num_threads(cores)
    if probe and n_envs > 4096:
        # bounded sample: probe at 4096 envs, then take the largest power-of-two shard (<= n_envs) whose ~8 steps fit
        small = cpu_baseline(spec, 4096, seed, target_seconds=2.0, probe=False, threads=cores)
        per_env = small["ms_per_step"] * 1e-3 / 4096
        fit = target_seconds / 8.0 / per_env
        n = 4096
        while n * 2 <= min(n_envs, fit):
            n *= 2
        n_envs = n

    from humanoid_amp_amd.robots import G1_JOINT_NAMES, G1_KEY_BODY_NAMES, HUMANOID_KEY_BODY_NAMES
    from humanoid_amp_amd.synthetic import make_state
    from humanoid_amp_amd.workloads import G1_REWARDS, make_disc_weights
    from humanoid_amp_amd.motions import MOTIONS_DIR
    from oracle import disc as odisc
    from oracle import env as oenv
    from oracle import motion as om

    from humanoid_amp_amd.workloads import clip_files

    mt = om.load_tables(clip_files(spec))
    g1 = spec.robot == "g1"
    perm = [mt.dof_names.index(n) for n in spec.joint_names] if g1 else list(range(len(mt.dof_names)))
    keys = [mt.body_names.index(n) for n in (G1_KEY_BODY_NAMES if g1 else HUMANOID_KEY_BODY_NAMES)]
    ref = mt.body_names.index(spec.reference_body)
    st = make_state(n_envs, spec.n_dof, spec.max_episode_length, mt.durations, seed, "cpu")
    times, ids = st["motion_times"].numpy(), st["motion_ids"].numpy()
    K, D = spec.K, spec.D
    weights = make_disc_weights(K * D, seed=0)
    mean, var = torch.zeros(K * D, dtype=torch.float64), torch.ones(K * D, dtype=torch.float64)
    buf = oenv.collect_reference(mt, times, ids, K, perm, ref, keys).view(n_envs, K, D).clone()
    lim = st["soft_limits"].unsqueeze(0).expand(n_envs, -1, -1)
    cfg = dict(G1_REWARDS)

    def step():
        with torch.no_grad():
            expert = oenv.collect_reference(mt, times, ids, K, perm, ref, keys)
            died, tout = oenv.dones(st["episode_length"], spec.max_episode_length, st["root_pos"][:, 2], 0.5)
            if g1:
                task, _ = oenv.g1_task_reward(cfg, st["root_lin_vel"], st["root_quat"], st["command"], died, st["actions"],
                                              st["joint_pos"], lim, st["joint_acc"], st["joint_vel"])
            else:
                task = torch.ones(n_envs)
            reset_ids = oenv.reset_env_ids(died, tout)
            obs = oenv.compute_obs(st["joint_pos"], st["joint_vel"], st["root_pos"], st["root_quat"], st["root_lin_vel"],
                                   st["root_ang_vel"], st["body_pos"])
            amp = oenv.shift_history(buf, obs)
            pol = oenv.actor_observation(obs, st["last_actions"], st["command"], use_command=True) if g1 else obs
            out = odisc.forward(weights, amp, mean, var, task=task.unsqueeze(-1), task_w=spec.task_weight,
                                style_w=spec.style_weight)
        return expert, reset_ids, pol, out

    step()  # warm-up (first touch, thread pool)
    t0 = time.perf_counter()
    step()
    one = time.perf_counter() - t0
    reps = int(max(min_reps, min(20, target_seconds / max(one, 1e-6))))
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    return {"value": n_envs / med, "unit": "env-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_envs} envs x {reps} steps of the oracle (torch-CPU restatement), median; {spec.name}",
            "ms_per_step": med * 1e3}


TRIVIAL_KERNEL_US = 3.4   # rocprofv3 duration of the bracketed one-workgroup command tick used as the probe: 3.37 us avg over 220 launches
                          # in profiles/r04_v_bench_kernel_trace_by_grid.md (command_kernel @ 1 workgroup) -- but 1.55 us avg (1.0-3.9) in
                          # r04_ai_*: a one-workgroup kernel's own duration moves by more than the correction, so the correction is
                          # reported (`us_corrected`) and NOT used for `achieved` / `frac`, which take the raw bracket (within 1 us of
                          # rocprofv3's average for the env launch in both profile sets: 52.1 vs 52.8, 54.0 vs 53.3)


def event_pair_overhead_us(device, reps=200):
    """What a HIP-event bracket of the engine's tracer adds to a launch: the median bracketed time of a ONE-workgroup kernel (the
    command-timer tick of 64 envs: 3.4 us under rocprofv3 when bracketed) minus that duration.  The tracer's per-kernel figures of bench.py are
    reported raw AND with this subtracted, so that they can be read next to a rocprofv3 kernel trace."""
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import command_step

    cmd, left = torch.zeros(64, 2, device=device), torch.full((64,), 1e9, device=device)
    for _ in range(20):
        command_step(cmd, left, mode=nat.AMP_COMMAND_TICK, step_dt=0.01, vel_range=(0.0, 1.0), time_range=(4.0, 7.0), seed=0, step=0)
    torch.cuda.synchronize()
    with nat.KernelTrace(capacity=reps + 8, kernel_filter="command_kernel") as tr:
        for _ in range(reps):
            command_step(cmd, left, mode=nat.AMP_COMMAND_TICK, step_dt=0.01, vel_range=(0.0, 1.0), time_range=(4.0, 7.0), seed=0, step=0)
        torch.cuda.synchronize()
    ms = sorted(m for _, m in tr.records())
    floor = ms[len(ms) // 2] * 1e3 if ms else 0.0
    return max(0.0, floor - TRIVIAL_KERNEL_US), floor


def dropin_env_step(spec, envs, device, steps=60, warmup=10):
    """The path skrl would drive: ``env.step(actions)`` of the drop-in env class (G1AmpEnv / HumanoidAmpEnv over the synthetic
    articulation, ``device_reset=True``: command timers -> DONES|REWARD launch -> one-launch device reset -> state-provider
    write -> OBS launch), timed eagerly and as a captured hipGraph (``env.capture_step()``), with the engine's kernels per
    step from the HIP-event tracer.  The toy physics of the synthetic articulation is switched off (closed PhysX step in the
    reference; scaffolding here), so the wall time is hooks + engine kernels + the provider's reset write."""
    import contextlib

    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.envs import G1AmpEnv, G1AmpEnvCfg_CUSTOM, HumanoidAmpEnv, HumanoidAmpEnvCfg
    from humanoid_amp_amd.motions import MOTIONS_DIR

    files = ",".join(os.path.join(MOTIONS_DIR, c + ".npz") for c in spec.clips)
    if spec.robot == "g1":
        cfg = G1AmpEnvCfg_CUSTOM(motion_file=files, num_amp_observations=spec.K, reset_strategy="random")
        cfg.decimation, cfg.episode_length_s = spec.decimation, spec.episode_length_s
        cls = G1AmpEnv
    else:
        cfg = HumanoidAmpEnvCfg(motion_file=files)
        cls = HumanoidAmpEnv
    cfg.scene.num_envs, cfg.sim.device = int(envs), str(device)
    robot = None
    if spec.robot != "g1":
        import numpy as np

        from humanoid_amp_amd.envs import SyntheticArticulation

        names = np.load(files.split(",")[0])
        robot = SyntheticArticulation(envs, names["dof_names"].tolist(), names["body_names"].tolist(), device, root_body="torso",
                                      dt=float(cfg.sim.dt), init_height=1.0)
    with contextlib.redirect_stdout(sys.stderr):
        env = cls(cfg, device_reset=True, reset_seed=0, robot=robot)
    env.robot.step = lambda: None
    env.reset()
    env.episode_length_buf.copy_(torch.randint(0, env.max_episode_length, (envs,), device=device))  # episode phases spread out
    acts = [torch.randn(envs, cfg.action_space, device=device) * 0.3 for _ in range(4)]

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            env.step(acts[i & 3])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    for i in range(warmup):
        env.step(acts[i & 3])
    eager = min(timed(steps), timed(steps))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with nat.KernelTrace(capacity=16 * 16) as tr:
        for i in range(16):
            env.step(acts[i & 3])
        torch.cuda.synchronize()
    traced = (time.perf_counter() - t0) / 16
    summ = tr.summary()
    per = {k: round(t / 16 * 1e3, 2) for k, (c, t) in summ.items()}
    brackets = sum(c for c, t in summ.values()) / 16.0   # event pairs per step
    env.capture_step()
    for i in range(warmup):
        env.step(acts[i & 3])
    graph = min(timed(steps), timed(steps))
    provider = per.get("scatter_rows_kernel", 0.0)
    faster = "eager" if eager <= graph else "hipgraph"
    out = {"envs": envs, "env_class": cls.__name__, "K": spec.K,
           "eager": {"us_per_step": eager * 1e6, "env_steps_per_s": envs / eager},
           "hipgraph": {"us_per_step": graph * 1e6, "env_steps_per_s": envs / graph},
           "recommended": faster,
           "recommended_note": "a graph replay pays ~8.5 us between replays (profiles/r02_small_shard_gaps.md) + the copy of the actions "
                               "into its static buffer; it wins where the eager step is HOST-bound (small shards) and loses where the "
                               "eager launches already queue back to back (GPU-bound: 65 536 envs)",
           "engine_kernel_us_per_step": per, "engine_env_side_us": round(sum(per.values()) - provider, 2),
           "traced_us_per_step": traced * 1e6, "event_pairs_per_step": brackets,
           "state_provider_reset_write_us": provider, "engine_launches_per_step": len(per) + (1 if "env_step_kernel" in per else 0),
           "note": "device_reset=True, no host sync inside step(); synthetic articulation with its toy physics off; kernel us are "
                   "HIP-event brackets of an eager traced pass (each carries `event_pair_overhead_us` of the top level)"}
    del env
    torch.cuda.empty_cache()
    return out


def measure_shard(spec, envs, device, rank, world, steps, warmup, use_graph, precision="f16x3", sets=0, seed=99):
    """Secondary measurement of one shard size: (env-steps/s whole-job, ms/step, launch mode, per-kernel us of an eager pass)."""
    import contextlib

    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.workloads import HotPath

    with contextlib.redirect_stdout(sys.stderr):
        hot = HotPath(spec, envs, device, seed=seed + rank, disc_precision=precision, state_sets=n_state_sets(spec, envs, sets))
    with nat.KernelTrace(capacity=16 * 8) as tr_all:
        for _ in range(8):
            hot.step()
    per_kernel = {k: round(t / 8 * 1e3, 2) for k, (c, t) in tr_all.summary().items()}
    launch, probes = choose_launch_mode(hot, envs, use_graph)
    settle(hot, max_seconds=2.0)
    # (a secondary point: the faster of two timed regions -- a host hiccup during an eager 50-step region has been seen to read 12 x slow)
    dts = min(timed_steps(hot, steps, warmup, world, None), timed_steps(hot, steps, 0, world, None))
    out = {"value": envs * world * steps / dts, "unit": "env-steps/s", "ms_per_step": dts / steps * 1e3, "envs_per_gpu": envs,
           "launch": launch, "launch_probes": probes, "kernel_us_per_step_eager": per_kernel, "state_sets": len(hot.states)}
    del hot
    torch.cuda.empty_cache()
    return out


def update_ms(marks_list):
    """Per-phase milliseconds of the updates whose HIP-event marks were collected (mean over the updates)."""
    if not marks_list:
        return {}
    acc = {}
    for marks in marks_list:
        marks[-1][1].synchronize()
        for k, (name, ev) in enumerate(marks[1:]):
            acc.setdefault(name, []).append(marks[k][1].elapsed_time(ev))
        acc.setdefault("total", []).append(marks[0][1].elapsed_time(marks[-1][1]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    import torch.distributed as dist

    # rehearsal knobs (CPU-side plumbing tests on a one-GPU box): AMP_BENCH_BACKEND=gloo, AMP_BENCH_DEVICE=0
    dev_index = int(os.environ.get("AMP_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("AMP_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm (xGMI inside a node)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import humanoid_amp_amd  # noqa: F401  (fails loudly if libamp_engine.so is missing)
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.distributed import shard_bounds
    from humanoid_amp_amd.workloads import WORKLOADS, AgentSide, HotPath, algorithmic_bytes_per_env_step, disc_flops_per_row

    import contextlib

    spec = WORKLOADS[args.workload]
    # ---- shard: strong scaling of the named global configuration unless --envs fixes the per-GPU size -------------
    if args.envs > 0:
        scaling, envs, global_envs = "weak", args.envs, args.envs * world
    else:
        scaling, global_envs = "strong", args.global_envs or DEFAULT_GLOBAL_ENVS[args.workload]
        lo, hi = shard_bounds(global_envs, world, rank)
        envs = hi - lo
    use_graph = args.graph and envs <= GRAPH_MAX_ENVS
    with contextlib.redirect_stdout(sys.stderr):  # MotionLoader prints like the reference; stdout carries only the JSON line
        hot = HotPath(spec, envs, device, seed=1234 + rank, disc_precision=args.disc_precision,
                      state_sets=n_state_sets(spec, envs, args.state_sets))
    dominant = DOMINANT_KERNEL[args.disc_precision]
    dominant_filter = DOMINANT_FILTER[args.disc_precision]

    # ---- the agent side: rollout store + discriminator update, the consumer of the path's one collective ----
    # world > 1: one update every --rollouts steps INSIDE the timed region (each rank contributes batch / world rows of every group of
    # every training step, ONE all-gather per update, all replicas then take the same optimizer steps); world == 1: the headline
    # region is the hot path alone, as SURVEY 8d defines the metric's unit of work, and the update is timed right after it
    update_every = args.update_every if args.update_every >= 0 else (args.rollouts if world > 1 else 0)
    n_train = args.epochs * args.mini_batches
    agent, marks, after_step = None, [], None
    if not args.no_update:
        with contextlib.redirect_stdout(sys.stderr):
            agent = AgentSide(hot, rollouts=args.rollouts, batch_size=args.batch, learning_epochs=args.epochs,
                              mini_batches=args.mini_batches, seed=4321, group=dist.group.WORLD if world > 1 else None, rank=rank)
        agent.updater.time_phases = True
    if agent is not None and update_every > 0:
        def after_step(i):
            agent.record()
            if (i + 1) % update_every == 0:
                agent.update()
                if i >= 0:
                    marks.append(agent.updater._marks)

    # ---- timed region: exactly --steps steps ------------------------------------------------------------------------
    # eager shards: the dominant kernel is bracketed by HIP events on its stream INSIDE the timed region; graph-replayed
    # shards (<= GRAPH_MAX_ENVS envs) cannot carry the tracer's event pairs, so their dominant-kernel duration comes from
    # an eager traced pass right after the timed region (said so in roofline.timing)
    launch, launch_probes = choose_launch_mode(hot, envs, use_graph)
    use_graph = getattr(hot, "_graphs", None) is not None
    settle(hot)
    if use_graph or envs <= GRAPH_MAX_ENVS:
        # small shards: the two event records per traced launch cost the host and the queue several us per step -- a quarter
        # of a 60-us step (measured: 4 096 envs 0.067 ms traced vs 0.048 ms untraced) -- so the timed region runs untraced
        # and the dominant kernel is timed in an eager traced pass of the same steps right after it
        dt = timed_steps(hot, args.steps, args.warmup, world, after_step)
        hot._graphs = None  # eager launches for the traced pass
        trace_every = 1
        with nat.KernelTrace(capacity=8 * (args.steps + args.warmup) + 8, kernel_filter=dominant_filter) as tr:
            for _ in range(args.steps + args.warmup):
                hot.step()
        timing = "eager traced pass of the same steps right after the (untraced) timed region"
    else:
        # an event pair is a barrier packet on the queue (~4 us each, measured: every launch bracketed costs a 356-us step
        # 18 us): bracket every TRACE_EVERY-th discriminator GEMM launch (coprime with the 2 or 4 launches of a step, so
        # both layers and every chunk are visited) -- still live, inside the timed region, on the launching stream
        trace_every = TRACE_EVERY
        with nat.KernelTrace(capacity=8 * (args.steps + args.warmup) + 8, kernel_filter=dominant_filter, every=trace_every) as tr:
            dt = timed_steps(hot, args.steps, args.warmup, world, after_step)
        timing = f"HIP events around every {trace_every}th launch of it inside the timed region"
    value = global_envs * args.steps / dt
    n_sets = len(hot.states)
    upd_in_region = update_ms(marks)

    # ---- the update outside the region (world == 1 default): one rollout stored, two updates (the first finds the replay ring
    #      empty), the second timed -----------------------------------------------------------------------------------------------
    upd_outside = {}
    if agent is not None and update_every == 0:
        for rep in range(2):
            for _ in range(args.rollouts):
                hot.step()
                agent.record()
            agent.update()
        upd_outside = agent.updater.phase_ms()
    plan = hot.disc.plan_info(envs)

    # ---- per-kernel picture of one step (all kernels traced, eager; outside the timed region, clocks settled) -----
    hot._graphs = None
    with nat.KernelTrace(capacity=16 * 16) as tr_all:
        for _ in range(16):
            hot.step()
    summary_all = tr_all.summary()
    per_kernel = {k: round(t / 16 * 1e3, 2) for k, (c, t) in summary_all.items()}  # us per step (all launches of the kernel)
    allrecs = [r for r in tr.records() if is_dominant(r[0])]  # the fused two-layer kernel, or layer 2
    if allrecs:
        dominant = allrecs[-1][0]
    fused = FUSED_KERNEL in summary_all
    fused_rows = envs
    if fused and any(is_dominant(k) and k != FUSED_KERNEL for k in summary_all):
        # a shard that is not a whole number of rounds of 128-row tiles: its full rounds run the fused kernel, the rest the
        # column-split two-kernel plan; the roofline object describes the fused launch (rows from amp_disc_plan_info)
        dominant = FUSED_KERNEL
        allrecs = [r for r in allrecs if r[0] == FUSED_KERNEL]
        fused_rows = plan["fused_rows"]
    per_step = max(1, round(sum(c for k, (c, t) in summary_all.items() if (k == FUSED_KERNEL if fused else is_dominant(k))) / 16))
    if trace_every == 1:
        recs = allrecs[-args.steps * per_step:]                        # small shards: traced pass = warmup + steps
    else:
        recs = allrecs[len(allrecs) * args.warmup // (args.steps + args.warmup):]  # samples of the timed steps
    if not recs:
        recs = allrecs or [(k, t / c) for k, (c, t) in summary_all.items() if is_dominant(k)]
        timing += " (too few timed launches sampled: warm-up / post-region launches included)"
    gemm2_ms = sum(ms for _, ms in recs) / max(len(recs), 1)

    out, detail = None, {}
    sustained = None
    if rank == 0 and args.disc_precision == "f16x3" and not os.environ.get("AMP_BENCH_NO_CALIBRATION"):
        # what the matrix pipes of THIS device sustain on a bare fp16 MFMA stream (amp_calibrate_mfma_f16): MI355X is power-
        # limited, and on operands that change from one MFMA to the next the figure is well below the nominal peak
        tc, fc = nat.calibrate_mfma_f16(False, 256, device, with_clock=True)
        tr_, fr = nat.calibrate_mfma_f16(True, 256, device, with_clock=True)
        t16, f16c = nat.calibrate_mfma_f16(True, 256, device, with_clock=True, layer2_stream=True)
        sustained = {"constant_operands": tc, "random_operands": tr_, "core_clock_mhz_constant_operands": fc,
                     "core_clock_mhz_random_operands": fr, "random_operands_16x16x32_two_waves": t16,
                     "core_clock_mhz_random_operands_16x16x32_two_waves": f16c,
                     "note": "constant / random_operands: bare v_mfma_f32_32x32x16_f16 stream, one wave per SIMD; the dominant kernel "
                             "issues v_mfma_f32_16x16x32_f16 from two waves per SIMD: its ceiling is the 16x16x32 figure"}
    pair_us, pair_floor = event_pair_overhead_us(device) if rank == 0 else (0.0, 0.0)
    r3 = lambda x: None if x is None else float(f"{x:.4g}")  # noqa: E731  (four significant digits keep the line short)
    if rank == 0:
        # HBM traffic of the dominant kernel: PMC FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per launch, collected
        # in separate rocprofv3 --pmc passes of this same command (tools/collect_profiles.sh) and committed: STATIC
        # numbers of the profiled run, not re-measured here (traffic_source says so)
        traffic, hbm_traffic, tj = None, None, {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            tj = {}
        rows = (fused_rows if fused else envs) // per_step
        # keys are "kernel<template args>@workgroups" of one launch (tools/pmc_summary.py --traffic-json)
        if fused:
            kx = (spec.K * spec.D + 31) // 32
            wgs = (rows + 127) // 128
            for name in (f"{FUSED_KERNEL}<{kx}, true>", f"{FUSED_KERNEL}<{kx}, false>", f"{FUSED_KERNEL}<{kx}>"):  # raw-row / plane input
                traffic = traffic or (tj.get(f"{name}@{wgs}") or {}).get("hbm_bytes")
        elif args.disc_precision == "f16x3" and rows >= 24576:
            wg = ((rows + 255) // 256 * 2 + 7) // 8 * 8
            traffic = (tj.get(f"disc_gemm_f16_dma_kernel<1, 4, 2, 0>@{wg}") or tj.get(f"disc_gemm_f16_dma_kernel<1, 4, 2>@{wg}") or {}).get("hbm_bytes")
        if spec.K == 2 and envs >= 32768 and envs % 32 == 0:
            wg = envs // 32 + (envs * spec.K + 255) // 256  # 32-env tiles + 256-sample expert tiles of the fused launch
            hbm_traffic = (tj.get(f"env_step_dma_reference_kernel<32>@{wg}") or {}).get("hbm_bytes")
        n_dom = fused_rows if fused else envs
        flops2 = (2.0 * n_dom * 1024 * 512 + 2.0 * n_dom * 512) / per_step   # layer 2 + the fused 512 -> 1 dot, per launch
        if fused:  # both layers in the launch: + layer 1 (ALGORITHMIC K D, not the padded 192)
            flops2 += 2.0 * n_dom * spec.K * spec.D * 1024 / per_step
        achieved = flops2 / (gemm2_ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[args.disc_precision]
        nprod = MFMA_PER_PRODUCT[args.disc_precision]
        hbm_kernels = ("env_step_reference_kernel", "collect_reference_kernel", "env_step_kernel", "compact_scatter_kernel",
                       "step_tail_kernel")
        hbm_us = sum(per_kernel.get(k, 0.0) for k in hbm_kernels)
        alg_bytes = algorithmic_bytes_per_env_step(spec) * envs
        env_us = per_kernel.get("env_step_reference_kernel", 0.0)
        ms_step = dt / args.steps * 1e3
        # ---- the agent update, reported separately --------------------------------------------------------------------------
        upd = None
        if agent is not None:
            ph = upd_in_region or upd_outside
            ex = agent.updater.exchange
            upd = {"in_timed_region": bool(marks), "every_steps": update_every or None, "updates_timed": len(marks) or (1 if ph else 0),
                   "ms_per_update": r3(ph.get("total")), "train_steps": n_train, "ms_per_train_step": r3(ph["train"] / n_train) if ph.get("train") else None,
                   "global_minibatch": args.batch, "rows_per_rank": args.batch // world,
                   "rollout_rows_per_rank": args.rollouts * envs}
            if ex is not None:
                upd.update({"produce_ms": r3(ph.get("produce")), "exchange_ms": r3(ph.get("exchange")), "train_ms": r3(ph.get("train")),
                            "allgather_mb_per_rank": round(ex.bytes_per_rank / 1e6, 2)})
            if marks:
                spent = sum(update_ms([m])["total"] for m in marks)
                upd["ms_per_step_excluding_updates"] = r3((dt * 1e3 - spent) / args.steps)
                upd["value_excluding_updates"] = r3(global_envs * args.steps / (dt - spent * 1e-3))
        collective = "none (one rank)"
        if world > 1 and agent is not None:
            collective = (f"ONE all-gather per discriminator update (RCCL via torch.distributed): [{n_train} steps x 3 groups x "
                          f"{args.batch // world} rows, {spec.K * spec.D}] f32 per rank -> the same {n_train} global minibatches of {args.batch} rows on "
                          f"every rank; update every {update_every} steps inside the timed region")
        out = {
            "metric": "AMP obs+motion-sample+reward env-steps/s", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": DTYPE[args.disc_precision], "data": "synthetic",
            "config": {"workload": f"{spec.description}; {global_envs} envs = {envs}/GPU; synthetic joint states ({n_sets} input sets "
                                   f"round-robin); discriminator [{spec.K * spec.D},1024,512,1] seed-0 init; "
                                   + (f"G1 CUSTOM-family reward scales, policy obs {spec.D - 12 + spec.n_dof + 2}" if spec.robot == "g1" else "task reward 1"),
                       "baseline_config": BASELINE_CONFIG.get((spec.name, global_envs, world), "not a BASELINE.json configuration"),
                       "envs_per_gpu": envs, "global_envs": global_envs, "parallelism": f"env-shard x{world}",
                       "launch": launch.split(":")[0], "disc_plan": {k: plan[k] for k in ("plan_name", "fused_rows", "chunk_rows", "env_overrides")},
                       "collective": collective, "update": upd},
            # achieved = ALGORITHMIC FLOPs / launch time against the dense MFMA peak of the operand type the kernel
            # issues; the fp16-split engine executes 3 MFMA products per algorithmic one (frac_executed counts those)
            "roofline": {"bound": "mfma", "kernel": dominant, "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "avg_launch_ms": gemm2_ms, "launches_timed": len(recs), "launches_per_step": per_step, "timing": timing,
                         "rows_per_launch": n_dom // per_step, "flops_per_launch": flops2,
                         "mfma_products_per_flop": nprod, "frac_executed": r3(nprod * achieved / peak),
                         "vs_fp32_mfma_peak": r3(achieved / MFMA_F32_PEAK_TFLOPS),
                         "frac_executed_of_sustained": r3(nprod * achieved / sustained["random_operands_16x16x32_two_waves"]) if sustained else None,
                         # the HBM-bound launch of the step (env step + expert sample), raw HIP-event bracket of the traced pass
                         "env_launch": {"bound": "hbm", "us": env_us, "bytes": alg_bytes, "achieved_gbs": r3(alg_bytes / (env_us * 1e-6) / 1e9) if env_us else None,
                                        "peak_gbs": HBM_PEAK_GBS, "frac": r3(alg_bytes / (env_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if env_us else None,
                                        "traffic": hbm_traffic}},
        }
        detail = {"launch": launch, "launch_probes": launch_probes, "state_sets": n_sets, "disc_plan": plan,
                  "mfma_sustained_tflops": sustained, "kernel_us_per_step": per_kernel,
                  "kernel_us_per_step_note": "a separate, fully traced eager pass after the timed region: every launch carries a HIP-event pair",
                  "roofline_hbm": {"kernels": list(hbm_kernels), "us": hbm_us, "bytes_per_env_step": algorithmic_bytes_per_env_step(spec),
                                   "achieved_gbs": alg_bytes / (hbm_us * 1e-6) / 1e9 if hbm_us else None},
                  "event_pair_overhead_us": round(pair_us, 2), "event_pair_probe_us": round(pair_floor, 2),
                  "env_launch_us_corrected": round(env_us - pair_us, 2),
                  "traffic_source": "profiles/pmc_traffic.json (static: rocprofv3 --pmc passes of the profiled run)",
                  "disc_flops_per_env_step": disc_flops_per_row(spec.K * spec.D),
                  "update_phases_ms": upd_in_region or upd_outside}
    del agent
    del hot
    torch.cuda.empty_cache()

    # ---- secondary points of the metric: the shard sizes of the multi-GPU configs (8192) and of configs[1] (4096) ----
    if not args.no_secondary:
        for n_sec in (8192, 4096):
            if n_sec == envs:
                continue
            sec = measure_shard(spec, n_sec, device, rank, world, max(args.steps, 50), args.warmup, args.graph,
                                args.disc_precision, args.state_sets)
            if rank == 0:
                detail[f"envs_{n_sec}"] = sec
                k = sec["kernel_us_per_step_eager"]
                gemm_us = sum(v for name, v in k.items() if name.startswith("disc_") and "finalize" not in name)
                fl = disc_flops_per_row(spec.K * spec.D) * n_sec
                out["roofline"][f"envs_{n_sec}"] = {"value": r3(sec["value"]), "ms_per_step": r3(sec["ms_per_step"]),
                                                    "gemm_us": r3(gemm_us), "frac": r3(fl / (gemm_us * 1e-6) / 1e12 / peak) if gemm_us else None}

    # ---- the same step on the fp32-MFMA GEMM engine (exact fp32 fma chain), for comparison ---------------------------
    if not args.no_fp32_engine and args.disc_precision != "f32" and world == 1:
        with contextlib.redirect_stdout(sys.stderr):
            hot_m = HotPath(spec, envs, device, seed=1234 + rank, disc_precision="f32", state_sets=n_state_sets(spec, envs, args.state_sets))
        settle(hot_m)
        with nat.KernelTrace(capacity=args.steps + args.warmup + 8, kernel_filter=DOMINANT_KERNEL["f32"]) as trm:
            dtm = timed_steps(hot_m, args.steps, args.warmup, world, None)
        rm = trm.records()[-args.steps:]
        ms32 = sum(ms for _, ms in rm) / max(len(rm), 1)
        flops32 = 2.0 * envs * 1024 * 512 + 2.0 * envs * 512  # the fp32 engine runs the shard as one launch
        out["roofline"]["fp32_engine"] = {
            "value": r3(global_envs * args.steps / dtm), "ms_per_step": r3(dtm / args.steps * 1e3), "dtype": "f32",
            "kernel": DOMINANT_KERNEL["f32"], "achieved": r3(flops32 / (ms32 * 1e-3) / 1e12), "peak": MFMA_F32_PEAK_TFLOPS,
            "frac": r3(flops32 / (ms32 * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS), "avg_launch_ms": r3(ms32)}
        del hot_m

    # ---- the drop-in env classes stepped the way skrl drives them (hooks): detail only -----------------------------------
    if world == 1 and not args.no_dropin and not spec.drop_dofs:   # (the drop-in G1AmpEnv is the 29-DoF robot)
        detail["dropin_env_step"] = [dropin_env_step(spec, n_env, device) for n_env in dict.fromkeys((envs, 8192, 4096))]

    # ---- BASELINE.json configs[2] / configs[3], bounded (the default line times configs[4] / [1] above) ---------------
    if world == 1 and not args.no_configs:
        cfgs, brief = {}, {}
        for key, wl, n in (("configs[1] literal: synthetic 23-DoF G1 (D=71) 4096 envs", "g1_walk_23dof", 4096),
                           ("configs[2] g1_dance K=10 8192 envs", "g1_dance", 8192),
                           ("configs[3] humanoid3 32768 envs on one GPU", "humanoid3", 32768),
                           ("configs[3] humanoid3 8192-env shard", "humanoid3", 8192)):
            cfgs[key] = measure_shard(WORKLOADS[wl], n, device, rank, world, 50, args.warmup, args.graph, args.disc_precision,
                                      args.state_sets)
            cfgs[key]["workload"] = WORKLOADS[wl].description
            brief[key] = {"value": r3(cfgs[key]["value"]), "ms_per_step": r3(cfgs[key]["ms_per_step"])}
        if not args.no_dropin:
            # the reference's self-consistent G1 family (Custom / Deploy cfg: K = 10, g1_amp_env_cfg.py:81-141,160-206) through the hooks
            cfgs["dropin_env_step K = 10 (G1AmpEnvCfg_CUSTOM on G1_dance)"] = [dropin_env_step(WORKLOADS["g1_dance"], n, device)
                                                                               for n in (8192, 65536)]
        detail["baseline_configs"] = cfgs
        out["config"]["other_configs"] = brief

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            # torch.set_num_threads(os.cpu_count()) -- and, next to it, the one-GPU box's 16-core share and one
            # thread; `value` is the FASTEST of the multi-threaded runs (value_from says which)
            share = cpu_baseline(spec, args.cpu_envs or envs, seed=1234, target_seconds=8.0)
            # (on a one-GPU box the container owns a 16-core share of a 256-thread host: 256 threads oversubscribe it ~1000x
            #  slower -- measured 2.9e2 env-steps/s -- so this leg runs ONE small sample and is bounded to a few seconds)
            every = cpu_baseline(spec, 256, seed=1234, target_seconds=2.0, probe=False, threads=os.cpu_count() or 1, min_reps=1)
            one = cpu_baseline(spec, 4096, seed=1234, target_seconds=3.0, probe=False, threads=1)
            best = every if every["value"] > share["value"] else share
            out["cpu_baseline"] = {"value": r3(best["value"]), "unit": "env-steps/s", "cores": best["cores"], "kind": "port",
                                   "sample": best["sample"], "ms_per_step": r3(best["ms_per_step"]), "cpu_model": cpu_model(),
                                   "nproc": os.cpu_count(), "value_1_thread": r3(one["value"]),
                                   "value_all_threads": r3(every["value"]), "value_16_thread_share": r3(share["value"])}
            detail["cpu_baseline_legs"] = {"threads_all": every, "threads_share": share, "threads_1": one}
            out["speedup_vs_cpu_baseline"] = r3(out["value"] / best["value"])
        # everything a reader of profiles/ wants beside the line: one JSON object on stderr + a side file; the LAST stdout line is the
        # contract's line and stays under 4 KB (the driver's record keeps `config`, `roofline`, `cpu_baseline` whole)
        detail_path = None
        try:
            ddir = os.path.join(ROOT, "gpurun_out")
            os.makedirs(ddir, exist_ok=True)
            detail_path = os.path.join(ddir, "bench_detail.json")
            json.dump({"line": out, "detail": detail}, open(detail_path, "w"))
        except OSError:
            detail_path = None
        print("[bench detail] " + json.dumps(detail), file=sys.stderr, flush=True)
        out["detail"] = "gpurun_out/bench_detail.json + the '[bench detail]' stderr line" if detail_path else "the '[bench detail]' stderr line"
        line = json.dumps(out)
        if len(line) > 4000:
            print(f"[bench] WARNING: the line is {len(line)} bytes (> 4000)", file=sys.stderr)
        print(line, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
