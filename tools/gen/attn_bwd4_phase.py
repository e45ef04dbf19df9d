#!/usr/bin/env python3
"""Emits BWD4_ITER, the pinned body of one query tile of emu_attn_bwd4_kernel (hoisdf_amd/csrc/attention_emu_bwd4.hip): 120
v_mfma_f32_32x32x16_bf16, each followed by the other work that issues behind it and a sched_barrier.

  slots   0- 23  S(t)    = Q . K^T        A = Q row fragments (LDS), B = kf (AGPR)
  slots  24- 47  dP(t)   = dO . V^T       A = dO row fragments (LDS), B = vf (AGPR)
  slots  48- 71  dQ(t-1) = K^T . dS^T     A = ktf (AGPR), B = dS^T(t - 1) through transpose reads; small products first, x0 y0 last
  slots  72- 95  dV^T   += dO^T . Pd      A = dO^T fragments (transpose reads), B = pw
  slots  96-119  dK^T   += Q^T . dS       A = Q^T fragments (transpose reads), B = gw

One wave per SIMD: the wave's own non-MFMA instructions (4 cycles of issue each, PMC) hide under its MFMAs only while every slot
carries less than an MFMA's 32 cycles of them - so the units are placed by a load balancer (cost model below) inside the window
their data allows:
  P units (P = exp2(S - lse), dropout, three-way split -> pw)   after S is complete, before dV
  Q units (dS = Pd dP - P delta, split -> gw, dS^T rows -> LDS) after dP is complete, before dK
  dropout decisions of tile t + 1                                anywhere (placed late: the dK slots are otherwise empty)
  dQ(t - 2) partial sums -> HBM, staging of tile t + 2, loads of tile t + 3, this wave's dQ(t - 1) partial -> LDS
Units of one family are ordered STAGE-major over element quads so that neighbours are independent.
    python tools/gen/attn_bwd4_phase.py > hoisdf_amd/csrc/attn_bwd4_phase.inc"""
import sys
PROD = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]          # (A plane, B plane): small terms first
# S = Q . K^T is accumulated in the FORWARD's order (attn_fwd2_phase.py FWD2_S: PROD over (K plane, Q plane) with A = K; here A = Q, so
# the pairs are swapped): the recomputed scores are then bit-identical to the ones the forward's LSE was made from - where |S| is large
# (a saturated first-layer attention: 2^21) a last-bit difference of S is a factor 1.19 of P (round 6, as in attn_bwd4h_phase.py)
PRODS = [(y, x) for x, y in PROD]
CAP = 22
COST = {"FR": 6, "TR": 9, "HA": 14, "HB": 18, "HC": 22, "LQ": 6, "DL": 6, "PA": 34, "PB": 20, "PC": 14, "PD": 14, "PE": 10,
        "QA": 22, "QB": 14, "QC": 14, "QD": 10, "TW": 22, "STQ": 8, "STS": 10, "LDG": 8, "LDS_": 6, "XOL": 14, "XOS": 12, "XW": 10, "XOP": 20, "XOW": 4, "XSIG": 8}


def staged(stages, extra=None):
    out = []
    for batch in range(2):
        qs = (2 * batch, 2 * batch + 1)
        for st in stages:
            out += [(st, q) for q in qs]
        if extra:
            out += [(extra, q) for q in qs]
    return out


def main():
    mf = []
    work = {m: [] for m in range(120)}
    load = [0] * 120

    def fixed(slot, text, kind):
        work[slot].append(text)
        load[slot] += COST[kind]

    # ---- the MFMA stream and its fragment reads (group g + 1 is read behind MFMAs 1..3 of group g) ----
    for m in range(24):                                   # S
        j, k = divmod(m, 6)
        x, y = PRODS[k]
        mf.append("%s(s, fr[%d][%d], kf[%d][%d])" % ("MFMA_SP" if m else "MFMA_SP0", j & 1, x, j, y))
    for m in range(24):                                   # dP
        j, k = divmod(m, 6)
        x, y = PROD[k]
        mf.append("%s(dp, fr[%d][%d], vf[%d][%d])" % ("MFMA_SP" if m else "MFMA_SP0", j & 1, x, j, y))
    bsrc = lambda ks, y: f"fq0[{ks}]" if y == 0 else f"fr[{ks & 1}][{y}]"
    for ks in range(4):                                   # dQ: five small products per key step ...
        for k in range(5):
            x, y = PROD[k]
            mf.append("%s(dq, ktf[%d][%d], %s)" % ("MFMA_Q" if (ks or k) else "MFMA_Q0", ks, x, bsrc(ks, y)))
    for ks in range(4):                                   # ... then the four x0 y0
        mf.append("MFMA_Q(dq, ktf[%d][0], fq0[%d])" % (ks, ks))
    for acc, pl in (("dv", "PWF"), ("dk", "GWF")):        # dV, dK: group g = (query step jj = g >> 1, d half mt = g & 1)
        for m in range(24):
            g, k = divmod(m, 6)
            x, y = PROD[k]
            mf.append("MFMA_VK(%s[%d], fr[%d][%d], %s(%d, %d))" % (acc, g & 1, g & 1, x, pl, y, g >> 1))
    for j in range(1, 4):
        for p in range(3):
            fixed(6 * (j - 1) + 1 + p, f"fr[{j & 1}][{p}] = FRQ({p}, {j})", "FR")
            fixed(24 + 6 * (j - 1) + 1 + p, f"fr[{j & 1}][{p}] = FRD({p}, {j})", "FR")
    for p in range(3):
        fixed(19 + p, f"fr[0][{p}] = FRD({p}, 0)", "FR")                               # dP's first group behind S's last
    fixed(43, "fq0[0] = FRT(0, 0)", "TR"); fixed(44, "fr[0][1] = FRT(1, 0)", "TR"); fixed(45, "fr[0][2] = FRT(2, 0)", "TR")
    for ks in range(1, 4):
        b = 48 + 5 * (ks - 1)
        fixed(b + 1, f"fq0[{ks}] = FRT(0, {ks})", "TR"); fixed(b + 2, f"fr[{ks & 1}][1] = FRT(1, {ks})", "TR"); fixed(b + 3, f"fr[{ks & 1}][2] = FRT(2, {ks})", "TR")
    for p in range(3):
        fixed(68 + p, f"fr[0][{p}] = FRA(3 + {p}, 0, 0)", "TR")                        # dV's first group behind dQ's closing products
    for ph, plane0 in ((72, 3), (96, 0)):
        for g in range(1, 4):
            for p in range(3):
                fixed(ph + 6 * (g - 1) + 1 + p, f"fr[{g & 1}][{p}] = FRA({plane0} + {p}, {g >> 1}, {g & 1})", "TR")
    for p in range(3):
        fixed(91 + p, f"fr[0][{p}] = FRA({p}, 0, 0)", "TR")                            # dK's first group
        fixed(115 + p, f"fr[0][{p}] = FRQN({p}, 0)", "FR")                             # next tile's first S group

    # ---- floating units: (name, arg) chains with a window [first, last]; a unit never sits ahead of its predecessor in the chain ----
    where = {}

    def place(chain, first, last, gap=1, after=None):
        """units of a dependency chain in order: each goes to the least loaded slot of a short look-ahead window that starts behind
        its predecessor (gap = 1: a later slot; 0: the same slot is allowed) and shrinks so that the rest of the chain still fits.
        ``after``: (name, arg) -> unit of another chain that must sit in an EARLIER slot (shared scratch registers)."""
        prev = first - gap
        n = len(chain)
        for k, (name, arg) in enumerate(chain):
            lo = max(first, prev + gap)
            if after and after(name, arg) is not None:
                lo = max(lo, where[after(name, arg)] + 1)
            lo = min(lo, last)
            hi = min(last - gap * (n - 1 - k), lo + 6)          # leave one slot per remaining unit
            hi = max(hi, lo)
            s = min(range(lo, hi + 1), key=lambda t: (load[t] + COST[name] > CAP, load[t], t))
            work[s].append(f"{name}({arg})" if arg is not None else f"{name}()")
            load[s] += COST[name]
            where[(name, arg)] = s
            prev = s
        return prev

    # dQ(t - 2): exchange-tile reads and the wait for / fetch of the previous key block's running sum right behind the barrier, the
    # add + store ~40 MFMAs later (the L2 round trip is hidden), the publication of this wave's count ~50 later (store
    # acknowledgement; its vmcnt(0) also covers the tile loads issued in between, which have landed by then)
    place([("XOL", 0), ("XOL", 1), ("XOP", None)], 0, 2, gap=0)
    # (ONE chain: a staging register is stored before it is re-loaded, and STS() moves the load offset)
    place([("STQ", i) for i in range(6)] + [("STS", None)] + [("LDG", i) for i in range(6)] + [("LDS_", None)], 3, 38, gap=0)
    place([("XOW", None), ("XOS", 0), ("XOS", 1)], 40, 47, gap=0)
    place([("XSIG", None)], 92, 96)
    place([("LQ", g) for g in range(4)], 20, 25, gap=0)
    place(staged(["PA", "PB", "PC", "PD", "PE"]), 26, 70)                              # S complete at slot 23 (+2), pw before slot 72
    place([("DL", g) for g in range(4)], 44, 49, gap=0)
    # dP complete at slot 47 (+2), gw before slot 96; the Q units of a quad reuse the scratch registers (xx, ff) of its P units
    place(staged(["QA", "QB", "QC", "QD"], "TW"), 50, 94, after=lambda name, q: ("PE", q) if name == "QA" else None)
    place([("XW", g) for g in range(4)], 74, 90, gap=0)                                # dq complete at slot 71 (+2)
    place([(st, q) for st in ("HA", "HB", "HC") for q in range(4)], 92, 119)           # dropout decisions of the NEXT tile

    lines = ["#define BWD4_ITER()", "  do {"]
    for m in range(120):
        w = "; ".join(work[m])
        lines.append(f"    {mf[m]}; {w}; SB();" if w else f"    {mf[m]}; SB();")
    lines.append("  } while (0)")
    width = max(len(l) for l in lines) + 1
    print("// generated by tools/gen/attn_bwd4_phase.py - the pinned body of one query tile of emu_attn_bwd4_kernel (one MFMA + the work behind it)")
    print("\n".join(l.ljust(width) + "\\" for l in lines[:-1]) + "\n" + lines[-1])
    print("// modelled issue cycles of the other work per slot: max %d, mean %.1f, slots over %d: %d" % (max(load), sum(load) / 120.0, CAP, sum(1 for x in load if x > CAP)), file=sys.stderr)
    print(" ".join(str(x) for x in load), file=sys.stderr)


if __name__ == "__main__":
    main()
