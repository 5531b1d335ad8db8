"""RAD-A2C update at the metric's size, split: update_model (K13 passes) and the policy iterations (K11 passes + K12 / K15), wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = RadSearchVec(N, number_agents=1, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
ag = {0: RNNAgentPPO(id=0, steps_per_epoch=480, steps_per_episode=120, alpha=0.1, seed=2)}
col = RNNCollector(env, ag, 480, 120)
a = ag[0]
orig_model, orig_pol = a.update_model, a.update_rada2c
acc = {"model": 0.0, "policy": 0.0, "n": 0}
def timed_model(B, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig_model(B, **kw)
    torch.cuda.synchronize(); acc["model"] += time.perf_counter() - t0
    return r
def timed_pol(B, kk, **kw):
    t0 = time.perf_counter()
    r = orig_pol(B, kk, **kw)
    acc["policy"] += time.perf_counter() - t0; acc["n"] += 1
    return r
a.update_model, a.update_rada2c = timed_model, timed_pol
for it in range(int(os.environ.get("A2C_ITERS", "3"))):
    acc.update(model=0.0, policy=0.0, n=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    col.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
    col.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: collect {1e3 * (t1 - t0):.1f} ms, update {1e3 * (t2 - t1):.1f} ms = update_model {1e3 * acc['model']:.1f} ms + "
          f"{acc['n']} policy iterations {1e3 * acc['policy']:.1f} ms (host time inside update_rada2c, incl. its one read) + rest", flush=True)
    ms = torch.cuda.memory_stats()
    print(f"        device mallocs so far {ms.get('num_device_alloc', -1)}, frees {ms.get('num_device_free', -1)}, retries {ms.get('num_alloc_retries', -1)}, "
          f"reserved {ms.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB", flush=True)
