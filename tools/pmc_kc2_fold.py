#!/usr/bin/env python3
"""folds the csv files of tools/pmc_kc2.sh: mean counter value per (label, case, counter) over the profiled launches"""
import sys, os, csv, collections
d = sys.argv[1]
tab = collections.defaultdict(dict)
for f in sorted(os.listdir(d)):
    if not f.endswith(".csv"): continue
    label, case, _ = f.split(".", 2)
    rows = list(csv.DictReader(open(os.path.join(d, f))))
    acc = collections.defaultdict(list)
    for r in rows:
        if os.environ.get("PMC_KERNEL", "emu_kc") in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        tab[(case, k)][label] = sum(v) / len(v)
labels = sorted({l for v in tab.values() for l in v})
print(f"{'case':34s} {'counter':34s} " + " ".join(f"{l:>14s}" for l in labels))
for (case, k), v in sorted(tab.items()):
    print(f"{case:34s} {k:34s} " + " ".join(f"{v.get(l, float('nan')):14.4g}" for l in labels))
