"""Per-epoch collect / update wall times of the CNN trainer over a run (does an epoch get slower as the policy learns?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.maps import CNNCritic
from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
N, E = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 30
T, L, A = 480, 120, 4
torch.manual_seed(2)
env = RadSearchVec(N, number_agents=A, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
gc = CNNCritic().cuda(); gco = torch.optim.Adam(gc.parameters(), lr=1e-3)
ag = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, alpha=0.1) for i in range(A)}
col = CNNCollector(env, ag, T, L, True)
for ep in range(E):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = col.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
    res = col.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    if ep % 3 == 0 or ep == E - 1:
        print(f"epoch {ep:3d}: collect {t1 - t0:.3f} s  update {t2 - t1:.3f} s  done {float(st['DoneCount'][0]):6.0f}  stop {[r.stop_iteration for r in res.values()]}  "
              f"mem {torch.cuda.memory_allocated() / 1e9:.2f} GB reserved {torch.cuda.memory_reserved() / 1e9:.2f} GB", flush=True)
