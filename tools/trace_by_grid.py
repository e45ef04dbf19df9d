#!/usr/bin/env python3
"""Per (kernel, grid size) statistics of the timed steps of a rocprofv3 --kernel-trace CSV of bench.py (what tools/trace_stats.py does
per kernel name): which SHAPES of a kernel family take the time.  usage: trace_by_grid.py <kernel_trace.csv> <warmup> <steps> <name substring>"""
import csv, sys
from collections import defaultdict
path, warmup, steps, sub = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size") or r.get("Grid_Size_X") or "?", r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "?"))
rows.sort()
opt = [i for i, r in enumerate(rows) if "adamw_chunks_kernel" in r[2]]
per = len(opt) // (warmup + steps)
first = opt[per * warmup - 1] + 1
agg = defaultdict(lambda: [0, 0])
for s, e, n, g, w in rows[first:opt[-1] + 1]:
    if sub in n:
        k = (n.split("(")[0][-40:], g, w)
        agg[k][0] += 1; agg[k][1] += e - s
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-42s grid %9s wg %5s  x %5.1f/step  %8.3f ms/step  avg %8.1f us" % (k[0], k[1], k[2], a[0] / steps, a[1] / steps / 1e6, a[1] / a[0] / 1e3))
