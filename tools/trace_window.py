#!/usr/bin/env python3
"""Per-kernel statistics of the last `steps` steps of a rocprofv3 --kernel-trace CSV, steps delimited by a marker kernel that runs a
fixed number of times per step (inference configs have no optimizer launch: `lattice_count_kernel`, twice per eval step, is the
first hot-path kernel of a forward).  Also reports the device's idle time inside the window (wall span minus the union of the
kernel intervals): a host-bound step shows up there.
usage: trace_window.py <kernel_trace.csv> <marker substring> <markers per step> <steps> > stats.csv"""
import csv, sys
from collections import defaultdict


def main():
    path, marker, per, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    mk = [i for i, r in enumerate(rows) if marker in r[2]]
    assert len(mk) >= per * (steps + 1), (len(mk), per, steps)
    first = mk[-per * (steps + 1)]                # first marker of the (steps + 1)-th last step ...
    last = mk[-per] - 1                           # ... up to the dispatch before the last step's first marker
    win = rows[first:last + 1]
    agg = defaultdict(lambda: [0, 0, 10**18, 0])
    for s, e, n in win:
        a = agg[n]
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    total = sum(a[1] for a in agg.values())
    # union of the kernel intervals (two streams overlap)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = win[-1][1] - win[0][0]
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "CallsPerStep", "TotalNsPerStep", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, a[0] / steps, a[1] / steps, a[1] / a[0], round(100.0 * a[1] / total, 3), a[2], a[3]])
    print(f"# window: {steps} steps, {span / steps / 1e6:.3f} ms/step wall, {total / steps / 1e6:.3f} ms/step of kernel time, "
          f"{busy / steps / 1e6:.3f} ms/step with at least one kernel running, {(span - busy) / steps / 1e6:.3f} ms/step idle, "
          f"{len(win) / steps:.0f} launches/step", file=sys.stderr)


if __name__ == "__main__":
    main()
