"""Obstacle env timing: rs_step (8192 envs, 1-5 rectangles; 4096 x 4 agents) and the fused rollout of config 3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.ppo import FusedCollector, VecAgentPPO


def time_step(N, A, reps=200):
    env = RadSearchVec(N, number_agents=A, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
    env.reset()
    acts = [torch.randint(0, 8, (N, A), device="cuda").to(torch.int8) for _ in range(16)]
    for i in range(20):
        env.step(acts[i % 16])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        env.step(acts[i % 16])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"rs_step obstacles  8192 x 1: {time_step(8192, 1):8.1f} us per launch", flush=True)
print(f"rs_step obstacles  4096 x 4: {time_step(4096, 4):8.1f} us per launch", flush=True)
N, T, L = 8192, 480, 120
env = RadSearchVec(N, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
ag = {0: VecAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1)}
col = FusedCollector(env, ag, T, L)
col.collect(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); col.collect(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"config 3 collect (8192 envs x 480 steps, rollout + GAE): {min(ts)*1e3:.2f} ms = {min(ts)/T*1e6:.1f} us per lock-step", flush=True)
print("env error flags:", env.error_flags())
