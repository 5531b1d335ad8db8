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

if mode.startswith("resume"):
    # a data-parallel run resumed from per-rank files (train_PPO.save_resume / load): 3 epochs in one go against 2 epochs + a fresh set
    # of objects + load + 1 epoch, on every rank; `out` is a directory, rank 0 writes <out>/result.pt
    from radiation_ppo_amd.train import train_PPO                    # noqa: E402
    arch = mode.split("-")[1]
    A = 2 if arch == "cnn" else 1
    N = total_envs // world
    kw = dict(seed=7, number_of_agents=A, actor_critic_architecture=arch, global_critic_flag=(arch == "cnn"), steps_per_epoch=20,
              steps_per_episode=8, save_freq=1, ppo_kwargs=dict(train_pi_iters=2, train_v_iters=2, train_pfgru_iters=2, alpha=0.1))

    def make(name, epochs):
        env = RadSearchVec(N, number_agents=A, obstruction_count=2, enforce_grid_boundaries=True, seed=11, env_id_base=rank * N)
        return train_PPO(env=env, logger_kwargs=dict(output_dir=os.path.join(out, name)), total_epochs=epochs, **kw)

    def params(sim):
        mods = []
        for ag in sim.agents.values():
            mods += [ag.pi, ag.critic, ag.model] if arch == "cnn" else [ag.agent]
        return torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()])

    whole = make("whole", 3)
    whole.train()
    first = make("first", 2)
    first.train()
    if world > 1:
        dist.barrier()                                               # every rank's resume_rank<r>.pt is on disk
    second = make("second", 3)
    second.load(os.path.join(out, "first"))
    second.train()
    same = all(torch.equal(getattr(whole.collector.buf, k), getattr(second.collector.buf, k))
               for k in ("obs", "act", "rew", "val", "logp", "cut", "adv", "ret")) and torch.equal(params(whole), params(second))
    # the shards must not be copies of each other (what loading rank 0's env state on every rank produced)
    head = whole.collector.buf.obs[:, 0].reshape(-1)[:64].clone()
    flag = torch.tensor([1.0 if same else 0.0], device="cuda")
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        heads = [torch.empty_like(head) for _ in range(world)]
        dist.all_gather(heads, head)
        distinct = not torch.equal(heads[0], heads[1])
    else:
        distinct = True
    if rank == 0:
        torch.save({"equal": bool(flag.item() == 1.0), "distinct_shards": distinct, "epochs": second.epochs_done,
                    "files": sorted(os.listdir(os.path.join(out, "first", "0_agent")))}, os.path.join(out, "result.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(0)
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
