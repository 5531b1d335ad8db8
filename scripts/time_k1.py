import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for N, obst, A in ((4096, 0, 1), (1 << 20, 0, 1), (8192, 5, 1), (1 << 18, 5, 1), (4096, 0, 4)):
    env = RadSearchVec(N, number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True, seed=bench.SEED)
    r = bench.time_step_kernel(env, reps=100)
    print(f"N={N} obst={obst} A={A}: avg {r['avg_ms']*1e3:.1f} us  median {r['median_ms']*1e3:.1f} us  train {r['train_ms']*1e3:.1f} us  -> {N/(r['train_ms']*1e-3)/1e6:.0f} M env-steps/s", flush=True)
    del env
