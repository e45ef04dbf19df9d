#!/usr/bin/env python3
"""Where the device idles inside a step: the gaps (no kernel running on any stream) of the last `steps` steps of a rocprofv3
--kernel-trace CSV, summed per (kernel before the gap -> kernel after it), largest first.
usage: trace_gaps.py <kernel_trace.csv> <marker substring> <markers per step> <steps>"""
import csv, sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("hoisdf::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:60]


def main():
    path, marker, per, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    mk = [i for i, r in enumerate(rows) if marker in r[2]]
    first, last = mk[-per * (steps + 1)], mk[-per] - 1
    win = rows[first:last + 1]
    gaps = defaultdict(lambda: [0, 0])
    cur_e, cur_n, idle = None, None, 0
    hist = defaultdict(int)
    for s, e, n in win:
        if cur_e is not None and s > cur_e:
            g = s - cur_e
            idle += g
            a = gaps[(short(cur_n), short(n))]
            a[0] += 1; a[1] += g
            hist[min(int(g / 1000), 50)] += g
        if cur_e is None or e > cur_e:
            cur_e, cur_n = e, n
    print(f"idle {idle / steps / 1e6:.3f} ms/step in {sum(a[0] for a in gaps.values()) / steps:.0f} gaps/step")
    print("gap length histogram (us bucket: ms/step):", {k: round(v / steps / 1e6, 3) for k, v in sorted(hist.items())})
    for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{t / steps / 1e3:9.1f} us/step  x{c / steps:6.1f}  {a}  ->  {b}")


if __name__ == "__main__":
    main()
