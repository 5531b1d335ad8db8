import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
from radiation_ppo_amd.maps import CNNCritic
N, T, L, A = 256, 240, 120, 4
env = RadSearchVec(N, number_agents=A, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
gc = CNNCritic().cuda(); gco = torch.optim.Adam(gc.parameters(), lr=1e-3)
ag = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, train_pi_iters=6, train_v_iters=6) for i in range(A)}
col = CNNCollector(env, ag, T, L, True)
for it in range(2):
    t0 = time.perf_counter(); col.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
    col.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(it, "collect", t1 - t0, "update", t2 - t1, flush=True)
