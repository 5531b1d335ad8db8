#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into a short,
committable summary (kernel names truncated, top-N by total time)."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w") as f:
    f.write(f"# source: {src}\n# total GPU kernel time: {tot/1e6:.3f} ms over {len(rows)} distinct kernels\n")
    f.write("name,calls,total_ms,avg_us,pct,min_us,max_us\n")
    for r in rows[:top]:
        name = r["Name"].replace(",", ";")
        if len(name) > 110:
            name = name[:107] + "..."
        f.write(f"\"{name}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},"
                f"{float(r['Percentage']):.2f},{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f}\n")
    ours = [r for r in rows if r["Name"].startswith("void rs_") or "rs_" in r["Name"][:12]]
    f.write("# --- hand-written kernels of this repo ---\n")
    for r in ours:
        f.write(f"\"{r['Name'][:110]}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},"
                f"{float(r['Percentage']):.2f},{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f}\n")
