"""Does it learn?  train_PPO on config-2-like settings for a number of epochs; prints return / found-source counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.train import train_PPO
N, E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 60
obst = int(sys.argv[3]) if len(sys.argv) > 3 else 0
arch = sys.argv[4] if len(sys.argv) > 4 else "ff"            # "ff" (2x64 MLP) or "rnn" (RAD-A2C: GRU + PFGRU)
env = RadSearchVec(N, number_agents=1, obstruction_count=obst, enforce_grid_boundaries=True, seed=289714752)
sim = train_PPO(env=env, logger_kwargs={}, ppo_kwargs=dict(observation_space=11, steps_per_epoch=480, steps_per_episode=120,
                                                           number_of_agents=1, alpha=0.1),
                seed=2, number_of_agents=1, actor_critic_architecture=arch, global_critic_flag=False,
                steps_per_epoch=480, steps_per_episode=120, total_epochs=E)
step = max(1, E // 12)
for upto in list(range(step, E, step)) + [E]:               # train in slices so that progress is printed while the run is going
    sim.total_epochs = upto
    sim.train()
    r = sim.loggers[0].rows[-1]
    print(f"epoch {r['Epoch']:3d}  MeanEpRet {r['MeanEpRet']:8.3f}  EpLen {r['EpLen']:6.1f}  DoneCount {r['DoneCount']:7.0f}  "
          f"kl {r['kl_divergence']:.4f}  stop {r['stop_iteration']}  Entropy {r['Entropy']:.3f}  loss_predictor {r['loss_predictor']:.4f}  "
          f"LocLoss {r['LocLoss']:.1f}", flush=True)
rows = sim.loggers[0].rows
print("per-epoch PPOItersPerSec:", [round(r["PPOItersPerSec"], 1) for r in rows[:3]], "...", [round(r["PPOItersPerSec"], 1) for r in rows[-3:]],
      "collector:", type(sim.collector).__name__)
