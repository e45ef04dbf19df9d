#!/usr/bin/env python3
"""Fold the per-case PMC CSVs of tools/pmc_collect.sh into ONE json list keyed by (C entry, kernel, shape) - never merged across
shapes.  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KB, FETCH_SIZE counts 128-byte requests in
64-byte units on gfx950 (x2).
MFMA pipe: SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / kernel duration = "busy GHz" (cycles the matrix pipe of an average SIMD was busy per
nanosecond of the kernel: needs no clock estimate), mfma_busy_of_peak_clock = that / 2.4 GHz (the share of the datasheet pipe).
Effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration - round 6: the counter's window is wider than the dispatch's own start /
end timestamps (it also sees the neighbouring dispatches' tails), which read 2.6-2.8 GHz for some records in rounds 4-5: a value above
the part's 2.4 GHz is REFUSED (null + note), and mfma_busy (the share of the ACTIVE cycles) is only given with a valid clock.
usage: pmc_fold.py <dir> > out.json"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_case import CASES


def rows(path, kernel_sub):
    out = {}
    if not os.path.exists(path):
        return out
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if kernel_sub in r["Kernel_Name"]:
                out.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    d = sys.argv[1]
    res = []
    for case, (entry, ksub, shape) in CASES.items():
        f = rows(os.path.join(d, f"{case}.pass1.csv"), ksub)
        w = rows(os.path.join(d, f"{case}.pass2.csv"), ksub)
        b = rows(os.path.join(d, f"{case}.pass3.csv"), ksub)
        q = rows(os.path.join(d, f"{case}.pass4.csv"), ksub)
        if not (f or w or b):
            continue
        mean = lambda xs, i: sum(x[i] for x in xs) / len(xs)
        e = {"case": case, "entry": entry, "kernel": ksub, "shape": list(shape)}
        if "FETCH_SIZE" in f:
            e["fetch_bytes"] = mean(f["FETCH_SIZE"], 0) * 1024 * 2
        if "WRITE_SIZE" in w:
            e["write_bytes"] = mean(w["WRITE_SIZE"], 0) * 1024
        if "fetch_bytes" in e and "write_bytes" in e:
            e["hbm_bytes_per_launch"] = e["fetch_bytes"] + e["write_bytes"]
        if "GRBM_GUI_ACTIVE" in b and "SQ_VALU_MFMA_BUSY_CYCLES" in b:
            act = mean(b["GRBM_GUI_ACTIVE"], 0) / 8.0
            dur_ns = mean(b["GRBM_GUI_ACTIVE"], 1)
            busy = mean(b["SQ_VALU_MFMA_BUSY_CYCLES"], 0) / 1024.0
            e["duration_us_under_pmc"] = round(dur_ns / 1e3, 1)
            e["mfma_busy_ghz"] = round(busy / dur_ns, 4)
            e["mfma_busy_of_peak_clock"] = round(busy / dur_ns / 2.4, 4)
            clk = act / dur_ns
            if clk <= 2.45:
                e["effective_clock_ghz"] = round(clk, 3)
                e["mfma_busy"] = round(busy / act, 4)
            else:
                e["effective_clock_ghz"] = None
                e["clock_note"] = "GRBM_GUI_ACTIVE / 8 / duration = %.3f GHz > 2.4: the counter window exceeds the dispatch; refused" % clk
        if "SQ_WAVE_CYCLES" in q and "SQ_WAIT_ANY" in q:         # share of the wave cycles parked at s_waitcnt / s_barrier, issue-stalled, issuing
            wc = mean(q["SQ_WAVE_CYCLES"], 0)
            e["wave_wait_frac"] = round(mean(q["SQ_WAIT_ANY"], 0) / wc, 4)
            if "SQ_WAIT_INST_ANY" in q: e["wave_issue_stall_frac"] = round(mean(q["SQ_WAIT_INST_ANY"], 0) / wc, 4)
            if "SQ_ACTIVE_INST_ANY" in q: e["wave_issuing_frac"] = round(mean(q["SQ_ACTIVE_INST_ANY"], 0) / wc, 4)
        res.append(e)
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
