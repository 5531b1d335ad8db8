#!/usr/bin/env python3
"""Per-kernel SQ counter summary from a rocprofv3 --pmc csv pass (mean over launches of matching kernels)."""
import csv, glob, os, sys
from collections import defaultdict
d, pat = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for p in pat:
            if p in name:
                acc[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
for p, cs in acc.items():
    wc = sum(cs.get("SQ_WAVE_CYCLES", [0])) / max(len(cs.get("SQ_WAVE_CYCLES", [1])), 1)
    print(p)
    for k, v in sorted(cs.items()):
        m = sum(v) / len(v)
        print(f"   {k:28s} {m:16.0f}  {100 * m / wc if wc else 0:6.1f} % of WAVE_CYCLES  (x{len(v)})")
