import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
env = RadSearchVec(N, obstruction_count=5, enforce_grid_boundaries=True, seed=289714752)
env.reset()
acts = torch.randint(0, 9, (N, 1), device="cuda").to(torch.int8)
for _ in range(30):
    env.step(acts)
torch.cuda.synchronize()
print("done")
