#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter CSVs (FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes) into
profiles/r01_pmc_traffic.json: per hoisdf kernel, mean counter value per launch and HBM bytes with the
gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests in 64-B units -> x2; both
counters are in KB).  usage: pmc_summarize.py <fetch.csv> <write.csv> [out.json]"""
import csv, json, os, sys
from collections import defaultdict


def mean_by_kernel(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and row["Kernel_Name"].startswith(("hoisdf::", "void hoisdf::")):
                acc[row["Kernel_Name"].replace("void ", "").split("(")[0]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "r01_pmc_traffic.json")
    f, w = mean_by_kernel(fetch, "FETCH_SIZE"), mean_by_kernel(write, "WRITE_SIZE")
    res = json.load(open(out)) if os.path.exists(out) else {}
    for k in sorted(set(f) | set(w)):
        fb, wb = f.get(k, 0.0) * 1024 * 2, w.get(k, 0.0) * 1024
        res[k] = {"fetch_KB_raw": f.get(k, 0.0), "fetch_bytes": fb, "write_KB_raw": w.get(k, 0.0), "write_bytes": wb,
                  "hbm_bytes_per_launch": fb + wb}
        print(f"{k:70s} fetch {fb/1e6:9.1f} MB  write {wb/1e6:9.1f} MB")
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
