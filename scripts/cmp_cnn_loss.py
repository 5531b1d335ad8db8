"""CNN update with the actor-loss kernel vs the torch tail on the mini RAD-TEAM config of tests/_dp_worker.py (one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.maps import CNNCritic
from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
out = []
for use in (True, False):
    N, T, L, A = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 24, 8, 2
    torch.manual_seed(4321)
    env = RadSearchVec(N, number_agents=A, obstruction_count=2, enforce_grid_boundaries=True, seed=77)
    gc = CNNCritic().cuda()
    gco = torch.optim.Adam(gc.parameters(), lr=1e-3)
    agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, train_pi_iters=3, train_v_iters=3, actor_learning_rate=3e-3) for i in range(A)}
    for ag in agents.values():
        ag.use_loss_kernel = use
    col = CNNCollector(env, agents, T, L, True)
    col.collect()
    res = col.update()
    flat = torch.cat([p.detach().reshape(-1) for ag in agents.values() for p in ag.pi.parameters()])
    out.append((res[0].kl_divergence, res[0].loss_policy, res[0].Entropy, res[0].ClipFrac, res[0].stop_iteration, flat))
    print(use, out[-1][:5], flush=True)
print("max param diff", float((out[0][5] - out[1][5]).abs().max()))
