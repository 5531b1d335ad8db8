"""BASELINE config 4 at full size: 4096 envs x 4 agents, CNN actors + global critic, obstacles, T=480, L=120."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
from radiation_ppo_amd.maps import CNNCritic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
CHUNK = int(sys.argv[3]) if len(sys.argv) > 3 else 524288
os.makedirs("gpurun_out", exist_ok=True)
LOG = open("gpurun_out/c4.log", "a")
def say(*a):
    print(*a, flush=True); print(*a, file=LOG, flush=True)
T, L, A = 480, 120, 4
env = RadSearchVec(N, number_agents=A, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
gc = CNNCritic().cuda(); gco = torch.optim.Adam(gc.parameters(), lr=1e-3)
ag = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, chunk=CHUNK) for i in range(A)}
col = CNNCollector(env, ag, T, L, True)
for it in range(ITERS):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    col.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
    say("collect", it, round(t1 - t0, 2), "s")
    res = col.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    say(json.dumps({"iter": it, "N": N, "chunk": CHUNK, "collect_s": t1 - t0, "update_s": t2 - t1, "env_steps_per_s": N * T / (t2 - t0),
                      "stop": [res[a].stop_iteration for a in range(A)], "kl": res[0].kl_divergence,
                      "mem_GB": torch.cuda.max_memory_allocated() / 1e9}))
