#!/usr/bin/env python3
"""Histogram (by grid size) of the kernels whose name contains a pattern, timed steps only.  usage: trace_hist.py trace.csv warmup steps pattern"""
import csv, sys
from collections import defaultdict
path, warmup, steps, pat = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rows = []
with open(path, newline="") as f:
    rd = csv.DictReader(f)
    gk = [k for k in rd.fieldnames if k.startswith("Grid_Size")]
    for r in rd:
        g = 1
        for k in gk:
            g *= max(int(r[k]), 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], g))
rows.sort()
opt = [i for i, r in enumerate(rows) if "adamw_chunks_kernel" in r[2] or "FusedAdam" in r[2]]
per = len(opt) // (warmup + steps)
first, last = opt[per * warmup - 1] + 1, opt[-1]
agg = defaultdict(lambda: [0, 0])
for s, e, n, g in rows[first:last + 1]:
    if pat in n:
        agg[g][0] += 1; agg[g][1] += e - s
for g, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"grid {g:10d} x{c / steps:6.1f}/step {t / steps / 1e3:8.1f} us/step avg {t / c / 1e3:6.1f} us")
