"""Does RAD-TEAM learn?  train_PPO with the CNN architecture (4 agents, global critic, PFGRU channel, obstacles) on a reduced env
count; prints per-epoch team return / found sources.  python scripts/learn_check_cnn.py [envs] [epochs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.train import train_PPO
N, E = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 30
env = RadSearchVec(N, number_agents=4, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
sim = train_PPO(env=env, logger_kwargs={}, ppo_kwargs=dict(steps_per_epoch=480, steps_per_episode=120, number_of_agents=4, alpha=0.1),
                seed=2, number_of_agents=4, actor_critic_architecture="cnn", global_critic_flag=True,
                steps_per_epoch=480, steps_per_episode=120, total_epochs=E)
sim.train()
rows = sim.loggers[0].rows
for r in rows[::max(1, E // 10)] + [rows[-1]]:
    print(f"epoch {r['Epoch']:3d}  MeanEpRet {r['MeanEpRet']:8.3f}  EpLen {r['EpLen']:6.1f}  DoneCount {r['DoneCount']:7.0f}  "
          f"kl {r['kl_divergence']:.4f}  stop {r['stop_iteration']}  Entropy {r['Entropy']:.3f}  loss_critic {r['loss_critic']:.4f}", flush=True)
print("per-epoch PPOItersPerSec:", [round(r["PPOItersPerSec"], 2) for r in rows[:3]], "...", [round(r["PPOItersPerSec"], 2) for r in rows[-3:]])
