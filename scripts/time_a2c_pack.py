"""Where the RAD-A2C update's time outside the K13 passes and the policy loop goes: advantage normalisation and pack_episodes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.ppo import normalize_advantages
from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector, pack_episodes
N = 4096
env = RadSearchVec(N, number_agents=1, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
ag = {0: RNNAgentPPO(id=0, steps_per_epoch=480, steps_per_episode=120, alpha=0.1, seed=2)}
col = RNNCollector(env, ag, 480, 120)
col.collect(); col.collect()
torch.cuda.synchronize()
buf = col.buf
for rep in range(3):
    t0 = time.perf_counter()
    adv = normalize_advantages(buf.adv[:, :, 0]); torch.cuda.synchronize(); t1 = time.perf_counter()
    B = pack_episodes(buf.obs[:, :, 0], buf.act[:, :, 0], adv, buf.ret[:, :, 0], buf.logp[:, :, 0], buf.source_tar, buf.cut[:, :, 0], n_total=N,
                      seed=1, epoch=rep, sort_by_length=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    lh = B.lens.tolist(); t3 = time.perf_counter()
    print(f"normalize {1e3 * (t1 - t0):.2f} ms, pack_episodes {1e3 * (t2 - t1):.2f} ms, lens.tolist {1e3 * (t3 - t2):.2f} ms, E = {len(lh)}, L = {B.X.shape[0]}")
t0 = time.perf_counter(); col.collect(); torch.cuda.synchronize(); print(f"collect {1e3 * (time.perf_counter() - t0):.1f} ms")
