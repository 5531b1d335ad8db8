"""Micro-driver for profiling: repeated fused PPO gradient passes + rollout launches at the bench sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.ppo import FusedCollector, VecAgentPPO, FusedPPOGrad
N, T, L = 4096, 480, 120
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
env = RadSearchVec(N, enforce_grid_boundaries=True, seed=289714752)
ag = {0: VecAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1)}
col = FusedCollector(env, ag, T, L)
col.collect(); torch.cuda.synchronize()
t0 = time.perf_counter(); col.collect(); torch.cuda.synchronize(); t_roll = time.perf_counter() - t0
buf = col.buf
X = buf.obs[:, :, 0].reshape(-1, 11); act = buf.act.reshape(-1); adv = buf.adv.reshape(-1); ret = buf.ret.reshape(-1)
lpo = buf.logp.reshape(-1); w = torch.full_like(adv, 1.0 / adv.numel())
f = FusedPPOGrad(ag[0].agent)
f(X, act, adv, ret, lpo, w, 0.2, 0.1); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    f(X, act, adv, ret, lpo, w, 0.2, 0.1)
torch.cuda.synchronize()
print(f"rollout {t_roll*1e3:.2f} ms; grad pass {(time.perf_counter()-t0)/reps*1e3:.3f} ms (M={X.shape[0]})")
# env-step kernel at the two sizes bench.py reports
acts = torch.randint(0, 9, (N, 1), device="cuda").to(torch.int8)
for _ in range(5):
    env.step(acts)
big = RadSearchVec(1 << 20, enforce_grid_boundaries=True, seed=289714752)
big.reset()
ab = torch.randint(0, 9, (1 << 20, 1), device="cuda").to(torch.int8)
for _ in range(5):
    big.step(ab)
torch.cuda.synchronize()
