#!/usr/bin/env python3
"""bench.py -- headline benchmark of the radiation-search PPO hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--configs 3,4|none]

Headline workload (BASELINE.json configs[1]): single-agent RadSearch, 1 source, no obstructions, 4096 envs per
GPU, 2x64 MLP actor-critic, walls enforced, 480 steps/epoch, 120 steps/episode, synthetic spawns from
the Philox streams (seed 289714752 = the reference's robust_seed(2)).

One bench "step" = one full PPO iteration: a rollout of 480 lock-steps of all envs with the policy in
the loop (forward, sampling, env step, buffer write, resets), the GAE pass, and the complete PPO update
(<= 40 Adam steps with KL early stop).  Nothing is skipped inside the timed region.
    value        = env steps/s over the whole job  = K * 480 * (envs per GPU * N) / wall
    ms_per_step  = wall per PPO iteration (PPO iters/s = 1000 / ms_per_step)
Rank 0 prints ONE JSON line.  Besides the contract keys it carries
    roofline                  the dominant kernel of the iteration (fused PPO loss+gradient pass, MFMA bound)
    roofline_env_step[_large_n], roofline_gae   the HBM-class kernels (K1 at 4096 and 2^20 envs, K4)
    cpu_baseline              the oracle (a port of the reference's env) on the host cores
    configs                   BASELINE configs 3 and 4 at their stated sizes: one or two PPO iterations each with
                              ms_per_step, phase split and their own roofline figures (N = 1 only)
all timed in this run with HIP events on the launch stream.

Multi-GPU: `--gpus N` with no launcher in the environment starts its own N ranks (torch.distributed.run,
one process per GPU, RCCL) BEFORE touching the GPU and relays rank 0's line; under the driver's own
`python -m torch.distributed.run ... bench.py --gpus N` it reads RANK/LOCAL_RANK/WORLD_SIZE as usual.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 289714752
ENVS_PER_GPU = 4096
T_EPOCH, L_EPISODE = 480, 120
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak = f32 vector peak (MI355X_MICROARCH.md, Matrix cores table)
# algorithmic bytes per agent-step of K1 at A=1, no obstacles (DESIGN.md "K1 bytes"):
#   read : x,y 8 + sp,prev 16 + oob_count 4 + aflags 1 + src 8 + intensity,bkg 8 + iter,tstep,episode 12 + done 1 + action 1 = 59
#   write: obs 44 + reward 4 + team 4 + done 1 + info 7 + x,y 8 + sp,prev 16 + oob_count 4 + aflags 1 + done,iter,tstep 9 = 98
K1_BYTES_PER_AGENT_STEP = 157
# K4 (rs_gae): read rew 4, val 4, cut 1, last_val 4; write adv 4, ret 4
GAE_BYTES_PER_SAMPLE = 21
# K6 (rs_rollout): buffer row written per env-step: obs 44 + act 8 + logp 4 + val 4 + rew 4 + last_val 4 + cut 1 + source_tar 8
ROLLOUT_BYTES_PER_ENV_STEP = 77
# algorithmic FLOPs per sample of one PPO loss+gradient pass over the FF_core actor+critic:
#   forward  2*(11*64 + 64*64 + 64*8) + 2*(11*64 + 64*64 + 64*1)                     = 20 352  (20.4 kFLOP)
#   backward 2*(512+512+4096+4096+704) [actor dW3,dh2,dW2,dh1,dW1] + 2*(64+64+4096+4096+704) = 37 888
PPO_GRAD_FLOPS_PER_SAMPLE = 58240
# CNN trunk (conv3x3(Cin->8)-ReLU-pool-conv3x3(8->16)-ReLU) multiply-adds per 27x27 image, zero-padded borders counted:
#   forward: conv1 729*9*4*8 over the 4 DENSE planes (+ 2*9*8 stamp adds for the actor's two one-hot channels, round 3: those channels
#   are no longer convolved) + conv2 169*9*8*16;  backward (weight gradients + dP1): dW2 + dP1 = 2 * conv2, dW1 = conv1 / 4 (only the
#   arg-max pixel of each pool window carries gradient) (+ 2*72 gathers).  Actor 404 784 MAC forward (round 2 counted 509 616 for six
#   dense channels), critic 404 640.
def cnn_trunk_flops(cin: int, backward: bool) -> float:
    c1, c2 = 729 * 9 * 4 * 8 + (144 if cin == 6 else 0), 169 * 9 * 8 * 16
    return 2.0 * ((2 * c2 + c1 / 4) if backward else (c1 + c2))


# ------------------------------------------------------------------------------------------------ CPU baseline
def _py_worker(args):
    wid, n_envs, seconds = args[:3]
    obst, agents = (args[3], args[4]) if len(args) > 3 else (0, 1)
    import random
    from oracle.radsearch_oracle import PhiloxDraws, RadSearchOracle
    envs = [RadSearchOracle(PhiloxDraws(SEED, wid * n_envs + i), number_agents=agents, obstruction_count=obst,
                            enforce_grid_boundaries=True) for i in range(n_envs)]
    rnd = random.Random(wid)
    steps = 0
    t_in = [0] * n_envs
    t_epoch = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        t_epoch += 1
        for i, e in enumerate(envs):
            e.step({a: rnd.randrange(9) for a in range(agents)})
            t_in[i] += 1
            steps += 1
            if e.done or t_in[i] == L_EPISODE or t_epoch == T_EPOCH:
                if t_epoch == T_EPOCH:
                    e.epoch_end = True                     # a new obstacle layout per 480-step epoch (train.py:482-484)
                e.reset()
                t_in[i] = 0
        if t_epoch == T_EPOCH:
            t_epoch = 0
    return steps, time.perf_counter() - t0


def _c_lib():
    import ctypes as C
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)          # a no-op when the library is current
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "librs_oracle.so"))
    lib.rso_bench.restype = C.c_long
    lib.rso_bench.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_long, C.c_int]
    lib.rso_bench2.restype = C.c_long
    lib.rso_bench2.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int]
    return lib


def _c_worker(args):
    wid, n_envs, target = args[:3]
    obst, agents = (args[3], args[4]) if len(args) > 3 else (0, 1)
    lib = _c_lib()
    t0 = time.perf_counter()
    if obst == 0 and agents == 1:
        n = lib.rso_bench(SEED, wid * n_envs, n_envs, target, L_EPISODE)
    else:
        n = lib.rso_bench2(SEED, wid * n_envs, n_envs, target, L_EPISODE, T_EPOCH, obst, agents)
    if n < 0:
        raise RuntimeError("the C oracle's bench loop raised an environment error flag")
    return n, time.perf_counter() - t0


def cpu_baseline_obstacles(obst: int, agents: int, seconds: float = 5.0):
    """The configuration's own CPU figure: the C restatement of the env with obstructions (oracle/radsearch_oracle.c, pinned event by
    event to the Python oracle by tests/test_oracle_c.py; obstacle geometry = the exact-lattice restatement of the visilibity calls) on
    the host cores, the config's obstruction_count and agent count, uniform random actions, a fresh layout every 480 lock-steps.  The
    pure-Python oracle (the port closest to the reference's Python env.step) is timed beside it on the same cores."""
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 16))
    probe = _c_worker((0, 8, 100_000, obst, agents))
    rate = probe[0] / probe[1]
    target = max(int(rate * seconds), 10_000)
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_c_worker, [(w, 8, target, obst, agents) for w in range(cores)])
        py = pool.map(_py_worker, [(w, 8, min(seconds, 4.0), obst, agents) for w in range(cores)])
    return {"value": sum(r[0] for r in res) / max(r[1] for r in res), "unit": "env steps/s", "cores": cores, "kind": "port",
            "single_core_value": rate,
            "python_port_all_cores_value": sum(r[0] for r in py) / max(r[1] for r in py),
            "python_port_single_core_value": py[0][0] / py[0][1],
            "sample": f"{cores} procs x 8 envs, obstruction_count={obst}, {agents} agent(s), uniform random actions, {target} env steps "
                      f"(~{seconds:.0f} s) each, oracle/radsearch_oracle.c (-O2, scalar); env step + reset only, no policy; python port: "
                      f"oracle/radsearch_oracle.py, same envs, ~{min(seconds, 4.0):.0f} s per process"}


def cpu_baseline(seconds: float = 8.0):
    """The oracle on the host cores (kind "port"): the C restatement of the reference's RadSearch.step/reset
    (oracle/radsearch_oracle.c, pinned to the Python oracle and through it to the reference's golden vectors), one
    process per core, 16 obstacle-free single-agent envs each, uniform random actions, ~`seconds` of work per core.
    The pure-Python oracle (closest to the reference's own Python env.step) is timed on one core and on all cores."""
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 16))
    probe = _c_worker((0, 16, 2_000_000))
    rate = probe[0] / probe[1]
    target = int(rate * seconds)
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_c_worker, [(w, 16, target) for w in range(cores)])
        py_all = pool.map(_py_worker, [(w, 16, min(seconds, 4.0)) for w in range(cores)])
    total = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    py = _py_worker((0, 16, min(seconds, 3.0)))
    return {"value": total / wall, "unit": "env steps/s", "cores": cores, "kind": "port",
            "single_core_value": rate, "python_port_single_core_value": py[0] / py[1],
            "python_port_all_cores_value": sum(r[0] for r in py_all) / max(r[1] for r in py_all),
            "sample": f"{cores} procs x 16 obstacle-free single-agent envs, uniform random actions, {target} env steps "
                      f"(~{seconds:.0f} s) each, oracle/radsearch_oracle.c (-O2, scalar); env step + reset only, no policy; "
                      f"python port: oracle/radsearch_oracle.py, same envs, ~{min(seconds, 4.0):.0f} s per process"}


# ------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) as a CHILD job and
    relay rank 0's JSON line.  Nothing in this parent process has touched the GPU (no HIP call, no torch.cuda init)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print(f"bench.py: the {args.gpus}-rank job failed (exit code {proc.returncode}, {len(lines)} result lines)", file=sys.stderr)
        return proc.returncode or 1
    print(lines[0], flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ kernel timing helpers
def _ms(pairs):
    return [a.elapsed_time(b) for a, b in pairs]


def time_step_kernel(env, reps: int = 400):
    """Average duration of the env-step kernel (rs_step_kernel) at this N, HIP events on the launch stream."""
    import torch
    N, A = env.num_envs, env.number_agents
    acts = torch.randint(0, 9, (N, A), device=env.device).to(torch.int8)
    env.reset()
    for _ in range(20):
        env.step(acts)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        ev[i][0].record()
        env.step(acts)
        ev[i][1].record()
        if i % 97 == 96:
            env.reset()
    torch.cuda.synchronize()
    ds = sorted(_ms(ev))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        env.step(acts)
    t1.record()
    torch.cuda.synchronize()
    return {"avg_ms": sum(ds) / len(ds), "median_ms": ds[len(ds) // 2], "train_ms": t0.elapsed_time(t1) / reps}


def time_grad_pass(col, reps: int = 20):
    """Average duration of one fused PPO loss+gradient pass (rs_ppo_grad: actor kernel + critic kernel + slab
    reduce) over the epoch's batch, back to back, HIP events on the launch stream."""
    import torch
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


def grad_roofline(events, stop_iters, iters_per_update: int, samples: int, iso_ms=None, traffic=None):
    """roofline object of K7 from the HIP-event brackets of the timed region.  Launches after a KL early stop are
    microsecond no-ops: only the first `stop_iteration` brackets of each update enter the mean."""
    ds = _ms(events)
    kept = []
    for i, k in enumerate(stop_iters):
        kept += ds[i * iters_per_update: i * iters_per_update + k]
    if not kept:
        kept = [iso_ms] if iso_ms else []
    g_ms = sum(kept) / max(len(kept), 1)
    tfl = PPO_GRAD_FLOPS_PER_SAMPLE * samples / (g_ms * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": "rs_ppo_grad2_kernel<8> + rs_ppo_grad2_kernel<1> (+ rs_ppo_reduce_kernel)",
           "achieved": tfl, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / MFMA_F32_PEAK_TFLOPS,
           "traffic": traffic["grad_pass_bytes_per_launch"] if traffic else None,
           "traffic_source": traffic.get("source") if traffic else None,
           "flops_per_launch": PPO_GRAD_FLOPS_PER_SAMPLE * samples, "samples": samples, "avg_launch_ms": g_ms,
           "launches_timed": len(kept), "launches_skipped_after_kl_stop": len(ds) - len(kept),
           "timing": "HIP events on the launch stream around every rs_ppo_grad launch of the timed region",
           "dtype": "f32 (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)"}
    if iso_ms is not None:
        out["avg_launch_ms_isolated"] = iso_ms
    return out


def pmc_traffic():
    """HBM bytes per launch from the newest committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json): measured by
    scripts/pmc_traffic.py on the same command, NOT in this run (counters need their own rocprofv3 passes)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        d["source"] = f"committed profile profiles/{os.path.basename(files[-1])} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes), not measured in this run"
        return d
    except Exception:  # noqa: BLE001
        return None


def unit_traffic(kernel: str):
    """(HBM bytes per unit, source note) of a kernel from the committed counter passes (profiles/r*_pmc_traffic.json `per_unit`: measured
    at the drivers' sizes by scripts/pmc_r3.sh and scaled per image / env-step / particle-step), or (None, None)."""
    d = pmc_traffic()
    u = (d or {}).get("per_unit", {}).get(kernel)
    if not u:
        return None, None
    return float(u["hbm_bytes_per_unit"]), f"{d['source']}; {u['hbm_bytes_per_unit']:.0f} B per {u['unit']} x the units of this launch"


# ------------------------------------------------------------------------------------------------ configs 3 and 4
def run_config3(dev, iters: int = 2, warmup: int = 1, cpu: bool = True):
    """BASELINE config 3: single agent, 1 source + random obstructions (U{1..5} rectangles per env, resampled every
    epoch), 8192 envs, 2x64 MLP -- the same PPO iteration as the headline."""
    import torch
    from radiation_ppo_amd import _lib
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.ppo import FusedCollector, VecAgentPPO
    N, T, L = 8192, T_EPOCH, L_EPISODE
    torch.manual_seed(SEED % (2 ** 31))
    env = RadSearchVec(N, obstruction_count=-1, enforce_grid_boundaries=True, seed=SEED, device=dev)
    ag = {0: VecAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1, device=dev)}
    col = FusedCollector(env, ag, T, L)
    for _ in range(warmup):
        col.collect(); col.update()
    torch.cuda.synchronize()
    _lib.EVENTS = {}
    tc = tu = 0.0
    stops = []
    t0 = time.perf_counter()
    for _ in range(iters):
        a = time.perf_counter(); col.collect(); torch.cuda.synchronize(); b = time.perf_counter()
        res = col.update(); torch.cuda.synchronize(); c = time.perf_counter()
        tc += b - a; tu += c - b
        stops.append(res[0].stop_iteration)
    dt = time.perf_counter() - t0
    ev, _lib.EVENTS = _lib.EVENTS, None
    roll_ms = sum(_ms(ev["rs_rollout"])) / iters
    gae_ms = sum(_ms(ev["rs_gae"])) / iters
    roll_gbs = ROLLOUT_BYTES_PER_ENV_STEP * N * T / (roll_ms * 1e-3) / 1e9
    # K1<true> on its own at this size (the lane function the rollout runs), HIP events
    k1 = time_step_kernel(env, reps=200)
    # the same kernel with one obstacle layout shared by 64 envs (geom_group_size = 64), throughput for both
    env64 = RadSearchVec(N, obstruction_count=-1, enforce_grid_boundaries=True, seed=SEED, geom_group_size=64, device=dev)
    env64.reset()
    k1s = time_step_kernel(env64, reps=200)
    del env64
    tr_roll, src_roll = unit_traffic("rs_rollout16_kernel<true>")
    tr_step, src_step = unit_traffic("rs_step4_kernel")
    # the obstacle step also reads the env's rectangles (16 B each) and the cached geodesics source -> rectangle vertex (4 x 8 B each):
    # SURVEY section 8d "+ 16 O B rectangle reads + 4 4 O B cached distances" (float64 here: 32 O)
    mean_obs = float(env.state("num_obs").float().mean().item())
    step4_bytes = K1_BYTES_PER_AGENT_STEP + (16 + 32) * mean_obs
    out = {"workload": "single-agent RadSearch, 1 source + U{1..5} random rectangles per env, 8192 envs, 2x64 MLP, 480 steps/epoch",
           "envs": N, "steps": iters, "warmup": warmup, "value": iters * N * T / dt, "unit": "env steps/s",
           "ms_per_step": 1e3 * dt / iters, "phase_ms": {"collect": 1e3 * tc / iters, "update": 1e3 * tu / iters},
           "update_adam_steps": stops,
           "roofline": grad_roofline(ev["rs_ppo_grad"], stops, ag[0].train_pi_iters, N * T),
           "config5_note": "config 5 (32 768 envs x 4 agents + obstacles on 8 GPUs) has config 4's per-GPU workload: 4096 envs x 4 agents",
           "roofline_rollout": {"bound": "hbm", "kernel": "rs_rollout16_kernel<true>", "achieved": roll_gbs, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": roll_gbs / HBM_PEAK_GBS, "traffic": None if tr_roll is None else tr_roll * N * T,
                                "traffic_source": src_roll, "avg_launch_ms": roll_ms,
                                "bytes_per_launch": ROLLOUT_BYTES_PER_ENV_STEP * N * T, "us_per_lock_step": 1e3 * roll_ms / T,
                                "note": "one launch = 480 lock-steps of 8192 envs with the policy in the loop; latency bound "
                                        "(f64 env chain + visibility tests per lane), not HBM bound"},
           "roofline_env_step": {"bound": "hbm", "kernel": "rs_step4_kernel (obstacles: four lanes per env)", "avg_launch_ms": k1["avg_ms"],
                                 "avg_launch_ms_shared_geometry_64": k1s["avg_ms"],
                                 "env_only_steps_per_s": N / (k1["avg_ms"] * 1e-3),
                                 "env_only_steps_per_s_shared_geometry_64": N / (k1s["avg_ms"] * 1e-3),
                                 "achieved": step4_bytes * N / (k1["avg_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": step4_bytes * N / (k1["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": None if tr_step is None else tr_step * N, "traffic_source": src_step,
                                 "bytes_per_launch": step4_bytes * N, "mean_rectangles_per_env": mean_obs,
                                 "note": f"{step4_bytes:.0f} B per env-step = 157 B state / outputs + (16 B rectangle + 32 B cached geodesics) x "
                                         f"{mean_obs:.2f} rectangles per env; latency bound (the serial exact-geometry chain of the slowest env of a wave)"},
           "gae_ms": gae_ms}
    flags = env.error_flags()
    out["env_error_flags"] = flags
    del col, env, ag
    torch.cuda.empty_cache()
    if cpu:
        out["cpu_baseline"] = cpu_baseline_obstacles(-1, 1)
    return out


def run_config4(dev, iters: int = 1, warmup: int = 1, cpu: bool = True):
    """BASELINE config 4: multi-agent RAD-TEAM, 4 agents, CNN actors + global critic on the heat maps, random
    obstructions, 4096 envs."""
    import torch
    from radiation_ppo_amd import _lib
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import CNNCritic
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
    N, T, L, A = 4096, T_EPOCH, L_EPISODE, 4
    torch.manual_seed(SEED % (2 ** 31))
    env = RadSearchVec(N, number_agents=A, obstruction_count=-1, enforce_grid_boundaries=True, seed=SEED, device=dev)
    gc = CNNCritic().to(dev)
    gco = torch.optim.Adam(gc.parameters(), lr=1e-3)
    ag = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, device=dev) for i in range(A)}
    col = CNNCollector(env, ag, T, L, True)
    for _ in range(warmup):
        col.collect(); col.update()
    torch.cuda.synchronize()
    _lib.EVENTS = {}
    tc = tu = 0.0
    stops = []
    t0 = time.perf_counter()
    for _ in range(iters):
        a = time.perf_counter(); col.collect(); torch.cuda.synchronize(); b = time.perf_counter()
        res = col.update(); torch.cuda.synchronize(); c = time.perf_counter()
        tc += b - a; tu += c - b
        stops.append([res[i].stop_iteration for i in range(A)])
    dt = time.perf_counter() - t0
    ev, _lib.EVENTS = _lib.EVENTS, None
    out = {"workload": "multi-agent RAD-TEAM, 4 agents, CNN actors + global critic, U{1..5} random rectangles, 4096 envs, "
                       "480 steps/epoch; step = 1 PPO iteration (rollout + GAE + 4 x <=40 actor + 40 critic Adam steps)",
           "envs": N, "agents": A, "steps": iters, "warmup": warmup, "value": iters * N * T / dt, "unit": "env steps/s",
           "ms_per_step": 1e3 * dt / iters, "phase_ms": {"collect": 1e3 * tc / iters, "update": 1e3 * tu / iters},
           "update_adam_steps": stops, "hbm_peak_allocated_GB": torch.cuda.max_memory_allocated(dev) / 1e9}
    # K9 / K10: images per launch = the update chunk; FLOP per image from the layer shapes
    chunk = min(ag[0].chunk, N * T)
    for name, key, bwd in (("rs_cnn_fwd_kernel (training forward)", "rs_cnn_trunk_forward_train", False),
                           ("rs_cnn_bwd_kernel", "rs_cnn_trunk_backward", True)):
        ds = _ms(ev.get(key, []))
        if ds:
            ms = sum(ds) / len(ds)
            # actor launches (6 channels) outnumber critic launches (4 channels) 4:1 with a global critic
            fl = (4 * cnn_trunk_flops(6, bwd) + cnn_trunk_flops(4, bwd)) / 5.0 * chunk
            tf = fl / (ms * 1e-3) / 1e12
            kn = "rs_cnn_bwd_kernel" if bwd else "rs_cnn_fwd_kernel"
            t6, src6 = unit_traffic(kn + "<6>")
            t4, _ = unit_traffic(kn + "<4>")
            out["roofline_cnn_bwd" if bwd else "roofline_cnn_fwd"] = {
                "bound": "mfma", "kernel": name, "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": None if t6 is None or t4 is None else (4 * t6 + t4) / 5.0 * chunk,
                "traffic_source": src6, "avg_launch_ms": ms, "launches_timed": len(ds),
                "images_per_launch": chunk, "flops_per_launch": fl,
                "note": "f32 vector/matrix peak (157.3 TFLOP/s); average over actor (6-channel) and critic (4-channel) launches"}
    out["env_error_flags"] = env.error_flags()
    del col, env, ag, gc, gco
    torch.cuda.empty_cache()
    if cpu:
        out["cpu_baseline"] = cpu_baseline_obstacles(-1, A)
    return out


def run_config_a2c(dev, iters: int = 2, warmup: int = 1, cpu: bool = True):
    """SURVEY section 8 row f2 (not a BASELINE config): the reference's original single-agent RAD-A2C -- GRU actor-critic + PFGRU
    location predictor, 15 PFGRU + <= 40 policy iterations over whole episodes per epoch."""
    import torch
    from radiation_ppo_amd import _lib
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector
    N, T, L = int(os.environ.get("RS_A2C_ENVS", ENVS_PER_GPU)), T_EPOCH, L_EPISODE     # the metric's size: 4096 envs
    torch.manual_seed(SEED % (2 ** 31))
    torch.cuda.reset_peak_memory_stats(dev)
    env = RadSearchVec(N, number_agents=1, obstruction_count=-1, enforce_grid_boundaries=True, seed=SEED, device=dev)
    ag = {0: RNNAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1, seed=2, device=dev)}
    col = RNNCollector(env, ag, T, L)
    for _ in range(warmup):
        col.collect(); col.update()
    torch.cuda.synchronize()
    _lib.EVENTS = {}
    tc = tu = 0.0
    stops, loops = [], []
    k13_units, k11_units = [], []                  # particle-steps of every timed K13 launch / K11 pass, in launch order (as the event pairs)
    t0 = time.perf_counter()
    for _ in range(iters):
        a = time.perf_counter(); col.collect(); torch.cuda.synchronize(); b = time.perf_counter()
        res = col.update(); torch.cuda.synchronize(); c = time.perf_counter()
        tc += b - a; tu += c - b
        k13_units += ag[0].k13_particle_steps; k11_units += ag[0].k11_particle_steps
        stops.append(res[0].stop_iteration)
        loops.append(1e3 * getattr(ag[0], "policy_loop_seconds", 0.0))
    dt = time.perf_counter() - t0
    ev, _lib.EVENTS = _lib.EVENTS, None
    out = {"workload": "single-agent RAD-A2C (GRU(13->24) actor-critic + PFGRU predictor, 40 particles), U{1..5} random rectangles, "
                       f"{N} envs, 480 steps/epoch; step = 1 PPO iteration (rollout + 15 PFGRU iterations (K13) + <=40 policy iterations "
                       "through K11 / K12)",
           "envs": N, "steps": iters, "warmup": warmup, "value": iters * N * T / dt, "unit": "env steps/s",
           "ms_per_step": 1e3 * dt / iters, "phase_ms": {"collect": 1e3 * tc / iters, "update": 1e3 * tu / iters},
           "policy_iterations": stops, "pfgru_iterations": ag[0].train_pfgru_iters, "env_error_flags": env.error_flags(),
           "hbm_peak_allocated_GB": torch.cuda.max_memory_allocated(dev) / 1e9}
    # where an update's time goes: the K13 passes (events), the policy loop's wall time, and the K11 passes of the policy iterations
    # as the GPU saw them (events on the side stream: ~13.7 ms each; a policy iteration cannot be shorter than its pass)
    kp = _ms(ev.get("rs_pfgru_pass", []))
    out["update_split_ms"] = {"k13_passes": sum(_ms(ev.get("rs_pfgru_train", []))) / iters, "policy_loop_wall": sum(loops) / max(len(loops), 1),
                              # (a pass call covers one policy iteration or, batched, several: per iteration = total / iterations)
                              "k11_pass_gpu_mean": (sum(kp) / max(sum(stops), 1)) if kp else None, "k11_passes_per_update": sum(stops) / iters,
                              "k11_pass_calls_per_update": len(kp) / iters}
    ds = _ms(ev.get("rs_pfgru_train", []))
    if ds:
        # K13: 9 561 multiply-adds per particle-step (forward cell 2 619, hid_obs forward + backward 1 248, transposed products 2 328,
        # weight-gradient outer products 3 366), 40 particles per episode-step.  Round 2 counted 12 180: it
        # included the forward cell RECOMPUTED in the backward walk, which round 3 replaced by stored gates -- not algorithmic work.
        ms = sum(ds) / len(ds)
        per_launch = k13_units[-len(ds):]                                # one count per launch, in launch order (as the event pairs)
        fl = 2.0 * 9561 * sum(per_launch) / len(per_launch)               # mean FLOP per launch; achieved = total FLOP / total time
        tf = fl / (ms * 1e-3) / 1e12
        t13, src13 = unit_traffic("rs_pfgru_train_kernel")
        out["roofline_pfgru_train"] = {"bound": "mfma", "kernel": "rs_pfgru_train_kernel (K13)", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                                       "traffic": None if t13 is None else t13 * sum(per_launch) / len(per_launch), "traffic_source": src13,
                                       "avg_launch_ms": ms,
                                       "launches_timed": len(ds), "particle_steps_per_launch": sum(per_launch) / len(per_launch),
                                       "flops_per_launch": fl,
                                       "note": "f32 vector/matrix peak (157.3 TFLOP/s); the forward walk's products run as scalar-weight FMAs on the "
                                               "VALU (measured f32 VALU peak 124 TFLOP/s), the backward walk's transposed products and "
                                               "weight-gradient reductions on the matrix cores"}
    if kp and k11_units:
        # K11 (rs_pfgru_kernel, four time steps per launch, ~30 launches per pass): 2 619 multiply-adds per particle-step -- the cell's
        # two gate products 27 -> 48, the observation-likelihood row 27 -> 1 (csrc/rs_pfgru.hip header); hid_obs runs once per SET and
        # step and is not counted.  Time = the HIP events around every rs_pfgru_pass call of the timed policy loops (side stream).
        n = min(len(kp), len(k11_units))
        fl = 2.0 * 2619 * sum(k11_units[-n:]) / n
        ms = sum(kp[-n:]) / n
        tf = fl / (ms * 1e-3) / 1e12
        t11, src11 = unit_traffic("rs_pfgru_kernel<false, 4>")          # per (episode, step) of the four-step launch
        out["roofline_pfgru_step"] = {"bound": "mfma", "kernel": "rs_pfgru_kernel (K11), per pass of the policy loop (rs_pfgru_pass)", "achieved": tf,
                                      "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                                      "traffic": None if t11 is None else t11 * sum(k11_units[-n:]) / n / 40.0, "traffic_source": src11,
                                      "avg_launch_ms": ms, "launches_timed": n, "particle_steps_per_launch": sum(k11_units[-n:]) / n,
                                      "flops_per_launch": fl,
                                      "note": "a 'launch' here is one PASS (reset + ~30 four-step launches over every episode of the epoch); f32 "
                                              "vector/matrix peak 157.3 TFLOP/s, the cell runs as scalar-weight FMAs on the VALU (measured f32 VALU "
                                              "peak 124 TFLOP/s); algorithmic FLOP, not issue slots"}
    del col, env, ag
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: 4096 envs per GPU (default); strong: 4096 envs in total (SURVEY 8d Metric 2)")
    ap.add_argument("--collector", choices=["fused", "torch"], default="fused")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for functional tests)")
    ap.add_argument("--configs", default=None, help="comma list of extra BASELINE configs to run at N=1 (default '3,4,a2c': BASELINE configs 3 and 4 and the RAD-A2C row f2; 'none')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(torch.cuda.device_count(), 1)      # functional multi-rank tests on a 1-GPU box share device 0
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    rccl_ranks = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local}"))
            one = torch.ones(1, device=f"cuda:{local}")
            dist.all_reduce(one)                               # an RCCL all-reduce over xGMI: counts the ranks that took part
            rccl_ranks = int(one.item())
        else:
            dist.init_process_group(backend=args.backend)
    else:
        torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    from radiation_ppo_amd import _lib
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.ppo import Collector, FusedCollector, VecAgentPPO

    if args.envs_per_gpu is not None:
        N = args.envs_per_gpu
    else:
        N = ENVS_PER_GPU if args.scaling == "weak" else ENVS_PER_GPU // world
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
    if rank == 0:
        _lib.EVENTS = {}                    # HIP events around the library launches of the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_iter(True)
    barrier()
    dt = time.perf_counter() - t0
    events, _lib.EVENTS = (_lib.EVENTS or {}), None
    dt_t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt = float(dt_t.item())

    env_steps = args.steps * T_EPOCH * N * world
    result = {
        "metric": "env steps/sec (whole node) at 4096 envs; PPO iters/sec at 1/2/4/8 GPUs",
        "value": env_steps / dt, "unit": "env steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32 policy / f64+int32 env", "data": "synthetic",
        "config": {"workload": f"single-agent RadSearch, 1 source, no obstructions, {N} envs/GPU, 2x64 MLP, "
                               "480 steps/epoch, 120 steps/episode, walls enforced; step = 1 PPO iteration "
                               "(rollout + GAE + full update)",
                   "envs_per_gpu": N, "envs_total": N * world, "steps_per_epoch": T_EPOCH, "steps_per_episode": L_EPISODE,
                   "collector": args.collector,
                   "parallelism": f"dp{world} (envs sharded by env_id_base, RCCL grad all-reduce per Adam step)"},
        "ppo_iters_per_s": args.steps / dt,
        "collector_env_steps_per_s": args.steps * T_EPOCH * N / max(phases["collect"], 1e-9) * world,
        "phase_ms": {k: 1000.0 * v / args.steps for k, v in phases.items()},
        "update_adam_steps": stop_iters,
        "backend": args.backend if world > 1 else None, "rccl_ranks": rccl_ranks,
    }

    if rank == 0:
        pmc = pmc_traffic() if N == ENVS_PER_GPU else None
        if args.collector == "fused":
            g_iso_ms, g_m = time_grad_pass(col)
            result["roofline"] = grad_roofline(events.get("rs_ppo_grad", []), stop_iters, agents[0].train_pi_iters, g_m,
                                               iso_ms=g_iso_ms, traffic=pmc)
            if events.get("rs_rollout"):
                r_ms = sum(_ms(events["rs_rollout"])) / len(events["rs_rollout"])
                r_gbs = ROLLOUT_BYTES_PER_ENV_STEP * N * T_EPOCH / (r_ms * 1e-3) / 1e9
                result["roofline_rollout"] = {"bound": "hbm", "kernel": "rs_rollout16_kernel<false>", "achieved": r_gbs,
                                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r_gbs / HBM_PEAK_GBS, "traffic": None,
                                              "avg_launch_ms": r_ms, "bytes_per_launch": ROLLOUT_BYTES_PER_ENV_STEP * N * T_EPOCH,
                                              "us_per_lock_step": 1e3 * r_ms / T_EPOCH,
                                              "note": "one launch = 480 lock-steps with the policy in the loop: latency bound"}
            if events.get("rs_gae"):
                g_ms = sum(_ms(events["rs_gae"])) / len(events["rs_gae"])
                g_gbs = GAE_BYTES_PER_SAMPLE * N * T_EPOCH / (g_ms * 1e-3) / 1e9
                result["roofline_gae"] = {"bound": "hbm", "kernel": "rs_gae_kernel", "achieved": g_gbs, "peak": HBM_PEAK_GBS,
                                          "unit": "GB/s", "frac": g_gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": g_ms,
                                          "bytes_per_launch": GAE_BYTES_PER_SAMPLE * N * T_EPOCH}
        k1 = time_step_kernel(env)
        bytes_per_launch = K1_BYTES_PER_AGENT_STEP * N * 1
        achieved = bytes_per_launch / (k1["avg_ms"] * 1e-3) / 1e9
        result["roofline_env_step"] = {"bound": "hbm", "kernel": "rs_step_kernel<false>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                       "traffic": pmc.get("env_step_4096_bytes_per_launch") if pmc else None,
                                       "traffic_source": pmc.get("source") if pmc else None,
                                       "bytes_per_launch": bytes_per_launch, "avg_launch_ms": k1["avg_ms"],
                                       "median_launch_ms": k1["median_ms"], "back_to_back_ms": k1["train_ms"],
                                       "env_only_steps_per_s": N / (k1["train_ms"] * 1e-3),
                                       "note": "4096 envs = 0.64 MB per launch: launch-latency bound, see roofline_env_step_large_n"}
        if world == 1:
            # the same kernel where it is bandwidth-sized: 2^20 envs
            try:
                big = RadSearchVec(1 << 20, number_agents=1, obstruction_count=0, enforce_grid_boundaries=True, seed=SEED, device=dev)
                kb = time_step_kernel(big, reps=60)
                bpl = K1_BYTES_PER_AGENT_STEP * (1 << 20)
                ach = bpl / (kb["avg_ms"] * 1e-3) / 1e9
                result["roofline_env_step_large_n"] = {"bound": "hbm", "kernel": "rs_step_kernel<false>", "envs": 1 << 20, "achieved": ach,
                                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                                       "traffic": pmc.get("env_step_1048576_bytes_per_launch") if pmc else None,
                                                       "traffic_source": pmc.get("source") if pmc else None,
                                                       "bytes_per_launch": bpl, "avg_launch_ms": kb["avg_ms"],
                                                       "env_only_steps_per_s": (1 << 20) / (kb["train_ms"] * 1e-3)}
                del big
            except Exception as e:  # noqa: BLE001
                result["roofline_env_step_large_n"] = {"error": repr(e)}
            which = args.configs if args.configs is not None else ("3,4,a2c" if N == ENVS_PER_GPU else "none")
            extra = {}
            del col, env
            torch.cuda.empty_cache()
            for c in [x.strip() for x in which.split(",") if x.strip() and x.strip() != "none"]:
                key = "row_f2_rada2c" if c == "a2c" else f"config{c}"
                try:
                    extra[key] = {"3": run_config3, "4": run_config4, "a2c": run_config_a2c}[c](dev, cpu=not args.no_cpu_baseline)
                except Exception as e:  # noqa: BLE001
                    extra[key] = {"error": repr(e)}
            if extra:
                result["configs"] = extra
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
