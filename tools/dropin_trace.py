"""For rocprofv3 --kernel-trace --stats: N steps of the drop-in env (device reset, physics off) and of HotPath at one size."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanoid_amp_amd.envs import G1AmpEnv, G1AmpEnvCfg_CUSTOM, G1AmpWalkEnvCfg
from humanoid_amp_amd.workloads import WORKLOADS, HotPath

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = 60
cfg = G1AmpEnvCfg_CUSTOM(motion_file=G1AmpWalkEnvCfg().motion_file, num_amp_observations=2, reset_strategy="random")
cfg.decimation, cfg.episode_length_s, cfg.scene.num_envs = 2, 10.0, n
with contextlib.redirect_stdout(io.StringIO()):
    env = G1AmpEnv(cfg, device_reset=True, reset_seed=0)
env.robot.step = lambda: None
env.reset()
env.episode_length_buf.copy_(torch.randint(0, env.max_episode_length, (n,), device="cuda"))
acts = [torch.randn(n, 29, device="cuda") * 0.3 for _ in range(4)]
for i in range(steps):
    env.step(acts[i & 3])
torch.cuda.synchronize()
with contextlib.redirect_stdout(io.StringIO()):
    hot = HotPath(WORKLOADS["g1_walk"], n, "cuda:0", seed=1, state_sets=3)
for _ in range(steps):
    hot.step()
torch.cuda.synchronize()
