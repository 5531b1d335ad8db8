"""Split of a RAD-A2C update at 1024 envs: PFGRU training passes (update_model) vs policy passes (update_rada2c)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.ppo import normalize_advantages
from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector, pack_episodes
N, T, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 480, 120
env = RadSearchVec(N, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
ag = {0: RNNAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1, seed=2)}
col = RNNCollector(env, ag, T, L)
col.collect(); col.update(); col.collect()
buf = col.buf
B = pack_episodes(buf.obs[:, :, 0], buf.act[:, :, 0], normalize_advantages(buf.adv[:, :, 0]), buf.ret[:, :, 0], buf.logp[:, :, 0], buf.source_tar,
                  buf.cut[:, :, 0], n_total=N, seed=1, epoch=1)
print("episodes", B.lens.shape[0], "longest", B.X.shape[0])
a = ag[0]
a.agent.train()
torch.cuda.synchronize(); t0 = time.perf_counter()
a.update_model(B); torch.cuda.synchronize(); t1 = time.perf_counter()
for it in range(10):
    a.update_rada2c(B, it)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"update_model ({a.train_pfgru_iters} iterations): {t1 - t0:.2f} s = {(t1 - t0) / a.train_pfgru_iters * 1e3:.0f} ms per iteration")
print(f"update_rada2c: {(t2 - t1) / 10 * 1e3:.1f} ms per iteration (x40 = {(t2 - t1) * 4:.2f} s)")
