#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace, csv output) of
`python3 scripts/prof_update.py N` into profiles/<name>.json: HBM bytes per launch per kernel.
Correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte fabric reads as 64 B on gfx950 -> x2;
WRITE_SIZE as is; unit KB."""
import csv, glob, json, os, sys
from collections import defaultdict

fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
driver_log = sys.argv[4] if len(sys.argv) > 4 else None          # the driver's stdout: its last JSON line names the units launched


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            if "rs_" not in name:
                continue
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            short += "@grid%s" % r["Grid_Size"]              # the same kernel is launched at several sizes: one entry per grid
            acc[short].append(float(r["Counter_Value"]))
    return acc


fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
kern = {}
for k in sorted(set(fe) | set(wr)):
    # steady state: drop the first launch of each kernel (cold caches), average the rest
    f = fe.get(k, [0.0]); w = wr.get(k, [0.0])
    f = f[1:] if len(f) > 1 else f
    w = w[1:] if len(w) > 1 else w
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    kern[k] = {"fetch_kb_raw": fk, "write_kb_raw": wk, "hbm_bytes_per_launch": int((2 * fk + wk) * 1024), "launches": len(f)}
res = {"_correction": "FETCH_SIZE x2 (gfx950 counts 128-B fabric reads as 64 B), WRITE_SIZE as is, KB -> x1024",
       "_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 <driver> (scripts/pmc_r3.sh)",
       "kernels": kern}
g8 = [v for k, v in kern.items() if "rs_ppo_grad2_kernel<8>" in k]
g1 = [v for k, v in kern.items() if "rs_ppo_grad2_kernel<1>" in k]
rd = [v for k, v in kern.items() if "rs_ppo_reduce_kernel" in k]
res["grad_pass_bytes_per_launch"] = sum(v["hbm_bytes_per_launch"] for v in g8 + g1 + rd)
for k, v in kern.items():
    if k == "rs_step_kernel<false>@grid4096":
        res["env_step_4096_bytes_per_launch"] = v["hbm_bytes_per_launch"]
    if k == "rs_step_kernel<false>@grid1048576":
        res["env_step_1048576_bytes_per_launch"] = v["hbm_bytes_per_launch"]
# the largest grid of every other kernel = the size the drivers (scripts/prof_update.py, scripts/prof_r3_kernels.py) mean to measure
big = {}
for k, v in kern.items():
    base, grid = k.split("@grid")
    if base not in big or int(grid) > big[base][0]:
        big[base] = (int(grid), v["hbm_bytes_per_launch"])
res["largest_grid_bytes_per_launch"] = {b: {"grid": g, "hbm_bytes_per_launch": t} for b, (g, t) in sorted(big.items())}
if driver_log and os.path.exists(driver_log):
    info = None
    for line in open(driver_log):
        if line.startswith('{"algorithmic_bytes_per_launch"'):
            info = json.loads(line)
    if info:
        lg = res["largest_grid_bytes_per_launch"]
        units = {"rs_cnn_fwd_kernel<6>": ("image", 32768), "rs_cnn_bwd_kernel<6>": ("image", 32768), "rs_cnn_fwd_kernel<4>": ("image", 32768),
                 "rs_cnn_bwd_kernel<4>": ("image", 32768), "rs_pfgru_kernel<false>": ("(owner, env) step", 4096 * 4),
                 "rs_pfgru_kernel<false, 1>": ("(owner, env) step", 4096 * 4), "rs_pfgru_kernel<false, 4>": ("(episode, step)", 16384 * 4),
                 "rs_pfgru_train_kernel": ("particle-step", info["k13_particle_steps"]),
                 "rs_rollout16_kernel<true>": ("env-step", 8192 * 480), "rs_step4_kernel": ("env-step", 8192)}
        alg = info["algorithmic_bytes_per_launch"]
        # one K13 pass = the forward walk + the backward walk (two launches since round 3): their traffic is reported together under the
        # backward kernel's name, which is what bench.py's roofline entry looks up
        fwd = [k for k in lg if k.startswith("rs_pfgru_train_fwd_kernel")]          # <true>: the draws hashed in the walk (the product path)
        if fwd and "rs_pfgru_train_kernel" in lg:
            fk = "rs_pfgru_train_fwd_kernel<true>" if "rs_pfgru_train_fwd_kernel<true>" in lg else fwd[0]
            lg["rs_pfgru_train_kernel"] = {"grid": lg["rs_pfgru_train_kernel"]["grid"],
                                           "hbm_bytes_per_launch": lg["rs_pfgru_train_kernel"]["hbm_bytes_per_launch"]
                                           + lg[fk]["hbm_bytes_per_launch"], "note": "forward (" + fk + ") + backward walk"}
        res["per_unit"] = {}
        for k, (unit, n) in units.items():
            if k in lg:
                a = alg.get(k, alg.get(k.split("<")[0]))
                if k == "rs_pfgru_kernel<false, 4>":
                    # its largest grid must be the full-size pass of the driver (2 731 workgroups of 256 threads), not a launch of the
                    # 1024-env update further down
                    assert lg[k]["grid"] == ((16384 + 5) // 6) * 256, lg[k]
                res["per_unit"][k] = {"unit": unit, "units_per_launch": n, "hbm_bytes_per_unit": lg[k]["hbm_bytes_per_launch"] / n,
                                      "algorithmic_bytes_per_unit": (a / n) if a else None,
                                      "traffic_over_algorithmic": (lg[k]["hbm_bytes_per_launch"] / a) if a else None}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in res.items() if not k.startswith("_") and k != "kernels"}))
for k, v in kern.items():
    print(f"{k:60s} fetch {2*v['fetch_kb_raw']/1024:9.2f} MB  write {v['write_kb_raw']/1024:9.2f} MB  x{v['launches']}")
