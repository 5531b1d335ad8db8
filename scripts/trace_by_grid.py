#!/usr/bin/env python3
"""Per-grid-size durations of one kernel from a rocprofv3 kernel_trace.csv: python scripts/trace_by_grid.py <trace.csv> <name substring> [<out>]
(K11's 120-step pass is 30 launches whose grids shrink with the number of episodes still running.)"""
import csv, sys, collections
src, pat = sys.argv[1], sys.argv[2]
by = collections.defaultdict(list)
for r in csv.DictReader(open(src)):
    if pat in r["Kernel_Name"]:
        by[int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
out.write(f"# {pat}: workgroups per launch -> launches, mean / min / max us, total ms\n")
tot = 0.0
for g in sorted(by, reverse=True):
    v = by[g]; tot += sum(v)
    out.write(f"{g:8d} wg  x{len(v):5d}  mean {sum(v) / len(v):9.1f}  min {min(v):9.1f}  max {max(v):9.1f}  total {sum(v) / 1e3:9.2f} ms\n")
out.write(f"# all launches: {tot / 1e3:.2f} ms\n")
