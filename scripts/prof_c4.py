"""Config 4 (4096 envs x 4 agents, CNN, global critic) with a shortened update (2 actor + 2 critic iterations) for rocprofv3:
python3 scripts/prof_c4.py [N] ; rocprofv3 --kernel-trace --stats -- python3 scripts/prof_c4.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.maps import CNNCritic
from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T, L, A = 480, 120, 4
env = RadSearchVec(N, number_agents=A, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
gc = CNNCritic().cuda(); gco = torch.optim.Adam(gc.parameters(), lr=1e-3)
ag = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, train_pi_iters=IT, train_v_iters=IT) for i in range(A)}
col = CNNCollector(env, ag, T, L, True)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    col.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
    col.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: collect {t1 - t0:.3f} s, update ({IT}+{IT} iterations) {t2 - t1:.3f} s -> per pass over {N * T} samples {(t2 - t1) / (IT * (A + 1)) * 1e3:.1f} ms", flush=True)
