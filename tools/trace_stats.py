#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED steps only, from a rocprofv3 --kernel-trace CSV of `bench.py`.

rocprofv3 --stats aggregates the whole process, including MIOpen's find-mode warm-up (which runs its
naive reference convolutions a few hundred times).  The timed region is delimited with the AdamW launches
(hoisdf::adamw_chunks_kernel, one per step): bench.py does `warmup` untimed steps, then `steps` timed ones, each ending with the same number of
optimizer kernels.  An inference trace (configs[3]/[4]: no optimizer) names another once-per-step kernel as the marker instead.
usage: trace_stats.py <kernel_trace.csv> <warmup> <steps> [marker substring] > stats.csv"""
import csv, sys
from collections import defaultdict


def main():
    path, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    markers = sys.argv[4:] or ["adamw_chunks_kernel", "FusedAdam"]
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    opt = [i for i, r in enumerate(rows) if any(m in r[2] for m in markers)]
    assert opt and len(opt) % (warmup + steps) == 0, (len(opt), warmup, steps)
    per = len(opt) // (warmup + steps)
    first = opt[per * warmup - 1] + 1            # first dispatch after the last warm-up optimizer kernel
    last = opt[-1]
    agg = defaultdict(lambda: [0, 0, 10**18, 0])
    for s, e, n in rows[first:last + 1]:
        a = agg[n]
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    total = sum(a[1] for a in agg.values())
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "CallsPerStep", "TotalNsPerStep", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, a[0] / steps, a[1] / steps, a[1] / a[0], round(100.0 * a[1] / total, 3), a[2], a[3]])
    span = rows[last][1] - rows[first][0]
    print(f"# timed window: {steps} steps, {span / steps / 1e6:.3f} ms/step wall, {total / steps / 1e6:.3f} ms/step of kernel time",
          file=sys.stderr)


if __name__ == "__main__":
    main()
