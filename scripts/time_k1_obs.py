import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
import bench
for N, obst in ((8192, 5), (262144, 5)):
    env = RadSearchVec(N, obstruction_count=obst, enforce_grid_boundaries=True, seed=bench.SEED)
    r = bench.time_step_kernel(env, reps=60)
    print(f"{os.environ.get('RS_LIB_PATH','default')[-14:]} N={N}: median {r['median_ms']*1e3:.1f} us", flush=True)
