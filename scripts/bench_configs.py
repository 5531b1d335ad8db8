"""Throughput of the BASELINE.json configs that are parity-test cases rather than the bench line (configs 3 and 4):
the same PPO iteration (rollout + GAE + full update) as bench.py, on their own sizes."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.ppo import FusedCollector, VecAgentPPO
from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
from radiation_ppo_amd.maps import CNNCritic

SEED = 289714752
out = {}

def run(col, iters, N, T):
    col.collect(); col.update(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    tc = 0.0
    for _ in range(iters):
        a = time.perf_counter(); col.collect(); torch.cuda.synchronize(); tc += time.perf_counter() - a
        res = col.update()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"last_stop_iteration": res[0].stop_iteration, "last_kl": res[0].kl_divergence,
            "env_steps_per_s": iters * N * T / dt, "ms_per_ppo_iter": 1e3 * dt / iters, "collect_ms": 1e3 * tc / iters}

# config 3: single agent, random obstructions, 8192 envs, 2x64 MLP
N, T, L = 8192, 480, 120
env = RadSearchVec(N, obstruction_count=-1, enforce_grid_boundaries=True, seed=SEED)
ag = {0: VecAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1)}
out["config3_obstacles_8192_mlp"] = run(FusedCollector(env, ag, T, L), 6, N, T)
del env, ag
if len(sys.argv) > 1 and sys.argv[1] == "c3only":
    print(json.dumps(out, indent=1)); sys.exit(0)
# config 4 (reduced: 256 envs): 4 agents, CNN, global critic, obstacles
N, T, L, A = 256, 120, 30, 4
env = RadSearchVec(N, number_agents=A, obstruction_count=-1, enforce_grid_boundaries=True, seed=SEED)
gc = CNNCritic().cuda(); gco = torch.optim.Adam(gc.parameters(), lr=1e-3)
ag = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, train_pi_iters=10, train_v_iters=10) for i in range(A)}
out["config4_reduced_256envs_4agents_cnn_10iters"] = run(CNNCollector(env, ag, T, L, True), 2, N, T)
print(json.dumps(out, indent=1))
