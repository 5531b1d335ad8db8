"""Worker for tests/test_bench_dist_gpu.py::test_data_parallel_equals_single_process: one PPO iteration (fused rollout +
GAE + device-side update) on this rank's shard of the envs; rank 0 writes the resulting parameters."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out, total_envs = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "ff"
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group(backend="gloo")
from radiation_ppo_amd.envs import RadSearchVec                      # noqa: E402
from radiation_ppo_amd.ppo import Collectives, FusedCollector, VecAgentPPO        # noqa: E402

if mode == "cnn":
    # BASELINE config 5 in miniature: multi-agent RAD-TEAM (CNN actors, global critic, obstacles) sharded over ranks
    from radiation_ppo_amd.maps import CNNCritic                      # noqa: E402
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector   # noqa: E402
    N, T, L, A = total_envs // world, 24, 8, 2
    torch.manual_seed(4321)
    env = RadSearchVec(N, number_agents=A, obstruction_count=2, enforce_grid_boundaries=True, seed=77, env_id_base=rank * N)
    gc = CNNCritic().cuda()
    gco = torch.optim.Adam(gc.parameters(), lr=1e-3)
    agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, train_pi_iters=3, train_v_iters=3,
                             actor_learning_rate=3e-3) for i in range(A)}
    for ag in agents.values():
        ag.sync_params()
        if os.environ.get("RS_TORCH_LOSS_TAIL"):                      # A/B switch: the torch composition of the actor loss
            ag.use_loss_kernel = False
    col = CNNCollector(env, agents, T, L, True)
    col.collect()
    c0 = Collectives.count
    res = col.update()
    if rank == 0:
        flat = torch.cat([p.detach().reshape(-1) for ag in agents.values() for p in ag.pi.parameters()]
                         + [p.detach().reshape(-1) for p in gc.parameters()]).cpu()
        # expected: per agent 2 (global advantage mean / std, mpi_statistics_scalar) + one per actor iteration; + one per critic iteration
        torch.save({"params": flat, "kl": res[0].kl_divergence, "loss": res[0].loss_policy, "stop": res[0].stop_iteration,
                    "entropy": res[0].Entropy, "loss_critic": res[0].loss_critic, "collectives": Collectives.count - c0,
                    "collectives_expected": sum(2 + res[i].stop_iteration for i in range(A)) + 3}, out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(0)
if mode == "rnn":
    # RAD-A2C (row f2): GRU actor-critic + PFGRU on the product kernels (K11 - K14), envs sharded over ranks
    from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector    # noqa: E402
    N, T, L = total_envs // world, 36, 12
    torch.manual_seed(99)
    env = RadSearchVec(N, number_agents=1, obstruction_count=2, enforce_grid_boundaries=True, seed=77, env_id_base=rank * N)
    agents = {0: RNNAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1, train_pi_iters=3, train_pfgru_iters=2, seed=5,
                             episode_chunk=16)}
    agents[0].sync_params()
    col = RNNCollector(env, agents, T, L)
    col.collect()
    c0 = Collectives.count
    res = col.update()[0]
    if rank == 0:
        flat = torch.cat([p.detach().reshape(-1) for p in agents[0].agent.parameters()]).cpu()
        # expected: 2 (advantage statistics) + one per PFGRU iteration (2) + one per policy iteration
        torch.save({"params": flat, "kl": res.kl_divergence, "loss": res.loss_policy, "stop": res.stop_iteration,
                    "entropy": res.Entropy, "loss_critic": res.loss_critic, "loss_predictor": res.loss_predictor,
                    "collectives": Collectives.count - c0, "collectives_expected": 2 + 2 + res.stop_iteration}, out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(0)
N, T, L = total_envs // world, 48, 12
torch.manual_seed(1234)                                              # same initial policy on every rank / world size
env = RadSearchVec(N, number_agents=1, obstruction_count=2, enforce_grid_boundaries=True, seed=77, env_id_base=rank * N)
# "ffstop": a tight KL target so that the early stop (ppo.py:1250-1261) triggers in the MIDDLE of the Adam loop -- the remaining
# iterations are no-ops that still all-reduce (zeros) on every rank
kl = 0.002 if mode == "ffstop" else 0.07
agents = {0: VecAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1, train_pi_iters=10 if mode == "ffstop" else 6,
                         actor_learning_rate=3e-3, target_kl=kl)}
agents[0].sync_params()
col = FusedCollector(env, agents, T, L)
col.collect()
res = col.update()[0]
if rank == 0:
    flat = torch.cat([p.detach().reshape(-1) for p in agents[0].agent.parameters()]).cpu()
    f = agents[0]._fused
    torch.save({"params": flat, "kl": res.kl_divergence, "loss": res.loss_policy, "stop": res.stop_iteration,
                "entropy": res.Entropy, "grads_after": f.grads.cpu(), "stats_after": f.stats.cpu()}, out)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
