import sys, torch
sys.path.insert(0, ".")
from humanoid_amp_amd.workloads import WORKLOADS, HotPath
hot = HotPath(WORKLOADS["g1_walk"], 8192, "cuda:0", seed=1, state_sets=8, two_streams=True)
for _ in range(30):
    hot.step()
hot.synchronize()
torch.cuda.synchronize()
