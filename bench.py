#!/usr/bin/env python3
"""bench.py -- headline benchmark of the radiation-search PPO hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): single-agent RadSearch, 1 source, no obstructions, 4096 envs per
GPU, 2x64 MLP actor-critic, walls enforced, 480 steps/epoch, 120 steps/episode, synthetic spawns from
the Philox streams (seed 289714752 = the reference's robust_seed(2)).

One bench "step" = one full PPO iteration: a rollout of 480 lock-steps of all envs with the policy in
the loop (forward, sampling, env step, buffer write, resets), the GAE pass, and the complete PPO update
(<= 40 Adam steps with KL early stop).  Nothing is skipped inside the timed region.
    value        = env steps/s over the whole job  = K * 480 * (4096 * N) / wall
    ms_per_step  = wall per PPO iteration (PPO iters/s = 1000 / ms_per_step)
Rank 0 prints ONE JSON line; it also carries `roofline` (the dominant kernel of the iteration: the fused PPO
loss+gradient pass, MFMA bound), `roofline_env_step` / `roofline_env_step_large_n` (the env-step kernel, HBM
class, at 4096 and 2^20 envs) -- all timed with HIP events in this run -- and `cpu_baseline` (the oracle, a
port of the reference's Python env, on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

SEED = 289714752
ENVS_PER_GPU = 4096
T_EPOCH, L_EPISODE = 480, 120
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
# algorithmic bytes per agent-step of K1 at A=1, no obstacles (DESIGN.md "K1 bytes"):
#   read : x,y 8 + sp,prev 16 + oob_count 4 + aflags 1 + src 8 + intensity,bkg 8 + iter,tstep,episode 12 + done 1 + action 1 = 59
#   write: obs 44 + reward 4 + team 4 + done 1 + info 7 + x,y 8 + sp,prev 16 + oob_count 4 + aflags 1 + done,iter,tstep 9 = 98
K1_BYTES_PER_AGENT_STEP = 157


def _py_worker(args):
    wid, n_envs, seconds = args
    import random
    from oracle.radsearch_oracle import PhiloxDraws, RadSearchOracle
    envs = [RadSearchOracle(PhiloxDraws(SEED, wid * n_envs + i), number_agents=1, obstruction_count=0,
                            enforce_grid_boundaries=True) for i in range(n_envs)]
    rnd = random.Random(wid)
    steps = 0
    t_in = [0] * n_envs
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for i, e in enumerate(envs):
            e.step({0: rnd.randrange(9)})
            t_in[i] += 1
            steps += 1
            if e.done or t_in[i] == L_EPISODE:
                e.reset()
                t_in[i] = 0
    return steps, time.perf_counter() - t0


def _c_lib():
    import ctypes as C
    import subprocess
    so = os.path.join(ROOT, "oracle", "_build", "librs_oracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(so)
    lib.rso_bench.restype = C.c_long
    lib.rso_bench.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_long, C.c_int]
    return lib


def _c_worker(args):
    wid, n_envs, target = args
    lib = _c_lib()
    t0 = time.perf_counter()
    n = lib.rso_bench(SEED, wid * n_envs, n_envs, target, L_EPISODE)
    return abs(n), time.perf_counter() - t0


def cpu_baseline(seconds: float = 10.0):
    """The oracle on the host cores (kind "port"): the C restatement of the reference's RadSearch.step/reset
    (oracle/radsearch_oracle.c, pinned to the Python oracle and through it to the reference's golden vectors), one
    process per core, 16 obstacle-free single-agent envs each, uniform random actions, ~`seconds` of work per core.
    The pure-Python oracle (closest to the reference's own Python env.step) is timed on one core beside it."""
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 16))
    probe = _c_worker((0, 16, 2_000_000))
    rate = probe[0] / probe[1]
    target = int(rate * seconds)
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_c_worker, [(w, 16, target) for w in range(cores)])
    total = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    py = _py_worker((0, 16, min(seconds, 4.0)))
    return {"value": total / wall, "unit": "env steps/s", "cores": cores, "kind": "port",
            "single_core_value": rate, "python_port_single_core_value": py[0] / py[1],
            "sample": f"{cores} procs x 16 obstacle-free single-agent envs, uniform random actions, {target} env steps "
                      f"(~{seconds:.0f} s) each, oracle/radsearch_oracle.c (-O2, scalar); env step + reset only, no policy"}


def time_step_kernel(env, reps: int = 400):
    """Average duration of the env-step kernel (rs_step_kernel) at this N, HIP events on the launch stream."""
    N, A = env.num_envs, env.number_agents
    acts = torch.randint(0, 9, (N, A), device=env.device).to(torch.int8)
    env.reset()
    for _ in range(20):
        env.step(acts)
    torch.cuda.synchronize()
    # per-launch brackets: event, launch, event -- each kernel is timed alone
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        ev[i][0].record()
        env.step(acts)
        ev[i][1].record()
        if i % 97 == 96:
            env.reset()
    torch.cuda.synchronize()
    ds = sorted(a.elapsed_time(b) for a, b in ev)
    # back-to-back train (kernel + launch gap), for reference
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        env.step(acts)
    t1.record()
    torch.cuda.synchronize()
    return {"avg_ms": sum(ds) / len(ds), "median_ms": ds[len(ds) // 2], "train_ms": t0.elapsed_time(t1) / reps}


MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak (MI355X_MICROARCH.md, Matrix cores table)
# algorithmic FLOPs per sample of one PPO loss+gradient pass over the FF_core actor+critic:
#   forward  2*(11*64 + 64*64 + 64*8) + 2*(11*64 + 64*64 + 64*1)                     = 20 352  (20.4 kFLOP)
#   backward 2*(512+512+4096+4096+704) [actor dW3,dh2,dW2,dh1,dW1] + 2*(64+64+4096+4096+704) = 37 888
PPO_GRAD_FLOPS_PER_SAMPLE = 58240


def time_grad_pass(col, reps: int = 20):
    """Average duration of one fused PPO loss+gradient pass (rs_ppo_grad: actor kernel + critic kernel + slab
    reduce) over the epoch's batch, HIP events on the launch stream."""
    from radiation_ppo_amd.ppo import FusedPPOGrad
    buf = col.buf
    ag = col.agents[0]
    X = buf.obs[:, :, 0].reshape(-1, 11)
    act, adv, ret, lpo = (t.reshape(-1) for t in (buf.act, buf.adv, buf.ret, buf.logp))
    w = torch.full_like(adv, 1.0 / adv.numel())
    f = ag._fused if getattr(ag, "_fused", None) is not None else FusedPPOGrad(ag.agent)
    f(X, act, adv, ret, lpo, w, 0.2, 0.1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f(X, act, adv, ret, lpo, w, 0.2, 0.1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, X.shape[0]


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json); None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)
    except Exception:  # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--collector", choices=["fused", "torch"], default="fused")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for functional tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(torch.cuda.device_count(), 1)      # functional multi-rank tests on a 1-GPU box share device 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend=args.backend)
    else:
        torch.cuda.set_device(local)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device(f"cuda:{local}")

    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.ppo import Collector, FusedCollector, VecAgentPPO

    N = args.envs_per_gpu
    torch.manual_seed(SEED % (2 ** 31))
    env = RadSearchVec(N, number_agents=1, obstruction_count=0, enforce_grid_boundaries=True, seed=SEED,
                       env_id_base=rank * N, device=dev)
    agents = {0: VecAgentPPO(id=0, steps_per_epoch=T_EPOCH, steps_per_episode=L_EPISODE, alpha=0.1, device=dev)}
    agents[0].sync_params()
    col = (FusedCollector if args.collector == "fused" else Collector)(env, agents, T_EPOCH, L_EPISODE)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    phases = {"collect": 0.0, "update": 0.0}
    stop_iters = []

    def one_iter(timed: bool):
        if timed:
            torch.cuda.synchronize(); a = time.perf_counter()
        col.collect()
        if timed:
            torch.cuda.synchronize(); b = time.perf_counter()
        res = col.update()
        if timed:
            torch.cuda.synchronize(); c = time.perf_counter()
            phases["collect"] += b - a
            phases["update"] += c - b
            stop_iters.append(res[0].stop_iteration)

    for _ in range(args.warmup):
        one_iter(False)
    fused = getattr(agents[0], "_fused", None)
    if fused is not None:
        fused.launch_events = []          # HIP events around every rs_ppo_grad launch of the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_iter(True)
    barrier()
    dt = time.perf_counter() - t0
    dt_t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt = float(dt_t.item())

    env_steps = args.steps * T_EPOCH * N * world
    result = {
        "metric": "env steps/sec (whole node) at 4096 envs; PPO iters/sec at 1/2/4/8 GPUs",
        "value": env_steps / dt, "unit": "env steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 policy / f64+int32 env", "data": "synthetic",
        "config": {"workload": "single-agent RadSearch, 1 source, no obstructions, 4096 envs/GPU, 2x64 MLP, "
                               "480 steps/epoch, 120 steps/episode, walls enforced; step = 1 PPO iteration "
                               "(rollout + GAE + full update)",
                   "envs_per_gpu": N, "steps_per_epoch": T_EPOCH, "steps_per_episode": L_EPISODE,
                   "collector": args.collector,
                   "parallelism": f"dp{world} (envs sharded, RCCL grad all-reduce)"},
        "ppo_iters_per_s": args.steps / dt,
        "collector_env_steps_per_s": args.steps * T_EPOCH * N / max(phases["collect"], 1e-9) * world,
        "phase_ms": {k: 1000.0 * v / args.steps for k, v in phases.items()},
        "update_adam_steps": stop_iters,
    }

    if rank == 0:
        pmc = pmc_traffic() if N == ENVS_PER_GPU else None
        # dominant kernel of the PPO iteration: the fused loss+gradient pass (<= 40 launches per iteration), MFMA bound
        if args.collector == "fused":
            evs = (fused.launch_events or []) if fused is not None else []
            if fused is not None:
                fused.launch_events = None
            g_iso_ms, g_m = time_grad_pass(col)
            g_ms = sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1) if evs else g_iso_ms
            tfl = PPO_GRAD_FLOPS_PER_SAMPLE * g_m / (g_ms * 1e-3) / 1e12
            result["roofline"] = {"bound": "mfma", "kernel": "rs_ppo_grad2_kernel<8> + rs_ppo_grad2_kernel<1> (+ rs_ppo_reduce_kernel)",
                                  "achieved": tfl, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / MFMA_F32_PEAK_TFLOPS,
                                  "traffic": pmc["grad_pass_bytes_per_launch"] if pmc else None,
                                  "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" if pmc else None,
                                  "flops_per_launch": PPO_GRAD_FLOPS_PER_SAMPLE * g_m, "samples": g_m,
                                  "avg_launch_ms": g_ms, "launches_timed": len(evs), "avg_launch_ms_isolated": g_iso_ms,
                                  "timing": "HIP events on the launch stream around every rs_ppo_grad launch of the timed region",
                                  "dtype": "f32 (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)"}
        k1 = time_step_kernel(env)
        bytes_per_launch = K1_BYTES_PER_AGENT_STEP * N * 1
        achieved = bytes_per_launch / (k1["avg_ms"] * 1e-3) / 1e9
        result["roofline_env_step"] = {"bound": "hbm", "kernel": "rs_step_kernel<false>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                              "traffic": pmc["env_step_4096_bytes_per_launch"] if pmc else None,
                              "bytes_per_launch": bytes_per_launch, "avg_launch_ms": k1["avg_ms"],
                              "median_launch_ms": k1["median_ms"], "back_to_back_ms": k1["train_ms"],
                              "env_only_steps_per_s": N / (k1["train_ms"] * 1e-3),
                              "note": "4096 envs = 0.64 MB per launch: launch-latency bound, see roofline_env_step_large_n"}
        # the same kernel where it is bandwidth-sized: 2^20 envs
        try:
            big = RadSearchVec(1 << 20, number_agents=1, obstruction_count=0, enforce_grid_boundaries=True, seed=SEED, device=dev)
            kb = time_step_kernel(big, reps=60)
            bpl = K1_BYTES_PER_AGENT_STEP * (1 << 20)
            ach = bpl / (kb["avg_ms"] * 1e-3) / 1e9
            result["roofline_env_step_large_n"] = {"bound": "hbm", "kernel": "rs_step_kernel<false>", "envs": 1 << 20, "achieved": ach,
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                          "traffic": pmc["env_step_1048576_bytes_per_launch"] if pmc else None, "bytes_per_launch": bpl,
                                          "avg_launch_ms": kb["avg_ms"], "env_only_steps_per_s": (1 << 20) / (kb["train_ms"] * 1e-3)}
            del big
        except Exception as e:  # noqa: BLE001
            result["roofline_env_step_large_n"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
