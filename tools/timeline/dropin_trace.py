"""A few drop-in env steps for a rocprofv3 --kernel-trace timeline: python tools/timeline/dropin_trace.py [envs] [K] [graph]"""
import contextlib, os, sys
import torch
sys.path.insert(0, ".")
from humanoid_amp_amd.envs import G1AmpEnv, G1AmpEnvCfg_CUSTOM
from humanoid_amp_amd.motions import MOTIONS_DIR

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
graph = len(sys.argv) > 3 and sys.argv[3] == "1"
clip = "G1_walk" if K == 2 else "G1_dance"
cfg = G1AmpEnvCfg_CUSTOM(motion_file=os.path.join(MOTIONS_DIR, clip + ".npz"), num_amp_observations=K, reset_strategy="random")
cfg.scene.num_envs, cfg.sim.device = envs, "cuda:0"
with contextlib.redirect_stdout(sys.stderr):
    env = G1AmpEnv(cfg, device_reset=True, reset_seed=0)
env.robot.step = lambda: None
env.reset()
env.episode_length_buf.copy_(torch.randint(0, env.max_episode_length, (envs,), device="cuda:0"))
acts = [torch.randn(envs, cfg.action_space, device="cuda:0") * 0.3 for _ in range(4)]
if graph:
    env.capture_step()
for i in range(24):
    env.step(acts[i & 3])
torch.cuda.synchronize()
