#!/usr/bin/env python3
"""Which kernels run beside the slow launches of one kernel?  python scripts/trace_overlap.py <kernel_trace.csv> <name substring> [<out>]
For every launch of the named kernel: its duration, and the time each other kernel overlaps it; launches are binned by slowdown over the
fastest one, and per bin the overlapping kernels' mean overlap per launch is listed."""
import csv, sys, collections, bisect
src, pat = sys.argv[1], sys.argv[2]
rows = []
for r in csv.DictReader(open(src)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [r[0] for r in rows]
mine = [r for r in rows if pat in r[2]]
tmin = min(e - s for s, e, _ in mine)
bins = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for s, e, _ in mine:
    slow = (e - s) / tmin
    b = 1.0 if slow < 1.15 else 1.3 if slow < 1.5 else 1.75 if slow < 2.0 else 2.5
    rec = bins[b]; rec[0] += 1; rec[1] += (e - s) / 1e3
    lo = bisect.bisect_left(starts, s - 5_000_000)
    for s2, e2, n2 in rows[lo:]:
        if s2 >= e:
            break
        if pat in n2:
            continue
        ov = min(e, e2) - max(s, s2)
        if ov > 0:
            rec[2][n2.split("(")[0][-60:]] += ov / 1e3
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
out.write(f"# {pat}: fastest launch {tmin / 1e3:.1f} us; bins by duration / fastest\n")
for b in sorted(bins):
    n, tot, c = bins[b]
    out.write(f"bin ~{b}x: {n} launches, mean {tot / n:.1f} us; kernels beside them (mean us of overlap per launch):\n")
    for k, v in c.most_common(8):
        out.write(f"      {v / n:8.1f}  {k}\n")
