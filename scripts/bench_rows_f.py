"""Measurement of the 'next' rows of SURVEY section 8 (not part of bench.py's line): RAD-A2C ('rnn') PPO iterations and the
Monte-Carlo evaluation harness, on one MI355X.  python scripts/bench_rows_f.py [envs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.evaluate import run_test_environments, sample_test_environments
from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T, L = 480, 120
torch.manual_seed(0)
env = RadSearchVec(N, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
ag = {0: RNNAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1, seed=2)}
col = RNNCollector(env, ag, T, L)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    col.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
    r = col.update()[0]; torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rnn iter {it}: {N} envs x {T} steps: collect {t1 - t0:.2f} s, update {t2 - t1:.2f} s "
          f"({ag[0].train_pfgru_iters} PFGRU + {r.stop_iteration} policy iterations) -> {N * T / (t2 - t0):.0f} env steps/s; "
          f"loss_predictor {r.loss_predictor:.4f} kl {r.kl_divergence:.5f} peak HBM {torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)
print("env error flags:", env.error_flags())
del col, env
torch.cuda.empty_cache()
sets = sample_test_environments(100, obstruction_count=3, seed=1)
for name, agent in (("rnn", ag[0]),):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res, summ = run_test_environments(agent, sets, montecarlo_runs=100, steps_per_episode=120, obstruction_count=3, seed=5)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"evaluation ({name}): 100 environments x 100 Monte-Carlo runs x <=120 steps in {dt:.2f} s = {10000 / dt:.0f} episodes/s; "
          f"success rate {summ['success_rate']:.3f}", flush=True)
