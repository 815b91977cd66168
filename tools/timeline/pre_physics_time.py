"""pre_physics_kernel brackets at 65 536 / 8 192 envs (G1 env, tick on): python tools/timeline/pre_physics_time.py"""
import contextlib, os, sys
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
from humanoid_amp_amd.envs import G1AmpEnv, G1AmpEnvCfg_CUSTOM
from humanoid_amp_amd.motions import MOTIONS_DIR
for envs in (65536, 8192):
    cfg = G1AmpEnvCfg_CUSTOM(motion_file=os.path.join(MOTIONS_DIR, "G1_walk.npz"), num_amp_observations=2, reset_strategy="random")
    cfg.scene.num_envs, cfg.sim.device = envs, "cuda:0"
    with contextlib.redirect_stdout(sys.stderr):
        env = G1AmpEnv(cfg, device_reset=True, reset_seed=0)
    env.reset()
    acts = [torch.randn(envs, cfg.action_space, device="cuda:0") * 0.3 for _ in range(4)]
    for i in range(20):
        env._pre_physics_step(acts[i & 3])
    torch.cuda.synchronize()
    with nat.KernelTrace(512, kernel_filter="pre_physics") as tr:
        for i in range(400):
            env._pre_physics_step(acts[i & 3])
        torch.cuda.synchronize()
    ms = sorted(m for _, m in tr.records())
    print(envs, "pre_physics_kernel bracket: median", round(ms[len(ms) // 2] * 1e3, 2), "us, p10", round(ms[len(ms) // 10] * 1e3, 2), flush=True)
