#!/usr/bin/env python3
"""Emits BWD4H_ITER, the pinned body of one query tile of emu_attn_bwd4h_kernel (hoisdf_amd/csrc/attention_emu_bwd4h.hip, the f16x2
form of the attention backward): 76 v_mfma_f32_32x32x16_f16, each followed by the other work that issues behind it and a sched_barrier.
Q, K, V, dO and Pd are two f16 pieces (three products per product); dS carries THREE pieces (its magnitude follows P and has no
useful a-priori bound: five products per product in the two contractions that read it).

  slots   0-11  S(t)    = Q . K^T        A = Q row fragments (LDS), B = kf (AGPR)
  slots  12-23  dP(t)   = dO . V^T       A = dO row fragments (LDS), B = vf (AGPR)
  slots  24-43  dQ(t-1) = K^T . dS^T     A = ktf (AGPR), B = dS^T(t - 1) through transpose reads (all twelve fragments fetched during
                                         dP); small products first, the four x0 y0 last
  slots  44-55  dV^T   += dO^T . Pd      A = dO^T fragments (transpose reads), B = pw
  slots  56-75  dK^T   += Q^T . dS       A = Q^T fragments (transpose reads), B = gw
The floating units (softmax, split, dropout decisions, exchange, staging) are placed by the load balancer of attn_bwd4_phase.py inside
the windows their data allows.
    python tools/gen/attn_bwd4h_phase.py > hoisdf_amd/csrc/attn_bwd4h_phase.inc"""
import os
import sys
P2 = [(1, 0), (0, 1), (0, 0)]                                    # (A plane, B plane) of two-piece operands: small terms first
# S = Q . K^T is accumulated in the FORWARD's order (attn_fwd2_phase.py FWD2_S2: K lo Q hi, K hi Q lo, K hi Q hi per k-step - here with
# A = Q, B = K: (Q hi, K lo) first): the recomputed scores are then BIT-IDENTICAL to the ones the forward's LSE was made from, so
# P = exp2(S - lse) cannot be thrown off by a last-bit difference of S where |S| is large (at |S| ~ 2^21 one ulp is a factor 1.19 of P)
P2S = [(0, 1), (1, 0), (0, 0)]
P23 = [(1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]                   # A two pieces, B (dS) three pieces
N = 76
CAP = int(os.environ.get("BWD4H_CAP", "34"))          # (A/B runs: the committed schedule is CAP = 34, look-ahead 4)
AHEAD = int(os.environ.get("BWD4H_AHEAD", "4"))
COST = {"FR": 6, "TR": 9, "HA": 14, "HB": 18, "HC": 30, "HD": 20, "LQ": 10, "DL": 10, "PA": 34, "PB": 20, "PC": 14, "PD": 10,
        "QA": 26, "QB": 14, "QC": 14, "QD": 10, "TW": 22, "STQ": 8, "STS": 10, "LDG": 8, "LDS_": 6, "XOL": 14, "XOS": 12, "XW": 10, "XOP": 20, "XOW": 4, "XSIG": 8}
S0, P0, Q0, V0, K0 = 0, 12, 24, 44, 56


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
    work = {m: [] for m in range(N)}
    load = [0] * N

    def fixed(slot, text, kind):
        work[slot].append(text)
        load[slot] += COST[kind]

    for m in range(12):                                   # S
        j, k = divmod(m, 3)
        x, y = P2S[k]
        mf.append("%s(s, fr[%d][%d], kf[%d][%d])" % ("MFMA_SP" if m else "MFMA_SP0", j & 1, x, j, y))
    for m in range(12):                                   # dP
        j, k = divmod(m, 3)
        x, y = P2[k]
        mf.append("%s(dp, fr[%d][%d], vf[%d][%d])" % ("MFMA_SP" if m else "MFMA_SP0", j & 1, x, j, y))
    for ks in range(4):                                   # dQ: four small products per key step ...
        for k in range(4):
            x, y = P23[k]
            mf.append("%s(dq, ktf[%d][%d], ft[%d][%d])" % ("MFMA_Q" if (ks or k) else "MFMA_Q0", ks, x, ks, y))
    for ks in range(4):                                   # ... then the four x0 y0
        mf.append("MFMA_Q(dq, ktf[%d][0], ft[%d][0])" % (ks, ks))
    for m in range(12):                                   # dV: group g = (query step jj = g >> 1, d half mt = g & 1)
        g, k = divmod(m, 3)
        x, y = P2[k]
        mf.append("MFMA_VK(dv[%d], fr[%d][%d], PWF(%d, %d))" % (g & 1, g & 1, x, y, g >> 1))
    for m in range(20):                                   # dK
        g, k = divmod(m, 5)
        x, y = P23[k]
        mf.append("MFMA_VK(dk[%d], fr[%d][%d], GWF(%d, %d))" % (g & 1, g & 1, x, y, g >> 1))
    assert len(mf) == N
    # fragment reads: group g + 1 behind the first MFMAs of group g
    for j in range(1, 4):
        for p in range(2):
            fixed(S0 + 3 * (j - 1) + 1 + p, f"fr[{j & 1}][{p}] = FRQ({p}, {j})", "FR")
            fixed(P0 + 3 * (j - 1) + 1 + p, f"fr[{j & 1}][{p}] = FRD({p}, {j})", "FR")
    for p in range(2):
        fixed(S0 + 9 + p, f"fr[0][{p}] = FRD({p}, 0)", "FR")                           # dP's first group behind S's last
    slot = P0                                                                        # the twelve dS^T(t - 1) fragments of dQ, one per dP slot
    for ks in range(4):
        for y in range(3):
            fixed(slot, f"ft[{ks}][{y}] = FRT({y}, {ks})", "TR")
            slot += 1
    for p in range(2):
        fixed(Q0 + 2 + p, f"fr[0][{p}] = FRA(2 + {p}, 0, 0)", "TR")                    # dV's first group (fr is idle during dQ)
    for g in range(1, 4):
        for p in range(2):
            fixed(V0 + 3 * (g - 1) + 1 + p, f"fr[{g & 1}][{p}] = FRA(2 + {p}, {g >> 1}, {g & 1})", "TR")
            fixed(K0 + 5 * (g - 1) + 1 + p, f"fr[{g & 1}][{p}] = FRA({p}, {g >> 1}, {g & 1})", "TR")
    for p in range(2):
        fixed(V0 + 9 + p, f"fr[0][{p}] = FRA({p}, 0, 0)", "TR")                        # dK's first group behind dV's last
        fixed(K0 + 16 + p, f"fr[0][{p}] = FRQN({p}, 0)", "FR")                         # next tile's first S group

    where = {}

    def place(chain, first, last, gap=1, after=None):
        prev = first - gap
        n = len(chain)
        for k, (name, arg) in enumerate(chain):
            lo = max(first, prev + gap)
            if after and after(name, arg) is not None:
                lo = max(lo, where[after(name, arg)] + 1)
            lo = min(lo, last)
            hi = min(last - gap * (n - 1 - k), lo + AHEAD)
            hi = max(hi, lo)
            s = min(range(lo, hi + 1), key=lambda t: (load[t] + COST[name] > CAP, load[t], t))
            work[s].append(f"{name}({arg})" if arg is not None else f"{name}()")
            load[s] += COST[name]
            where[(name, arg)] = s
            prev = s
        return prev

    place([("XOL", 0), ("XOL", 1), ("XOP", None)], 0, 2, gap=0)
    # (ONE chain: a staging register is stored before it is re-loaded, and STS() moves the load offset)
    place([("STQ", i) for i in range(4)] + [("STS", None)] + [("LDG", i) for i in range(4)] + [("LDS_", None)], 2, 22, gap=0)
    place([("XOW", None), ("XOS", 0), ("XOS", 1)], 24, 30, gap=0)
    place([("XSIG", None)], 56, 60)
    place([("LQ", g) for g in range(4)], 8, 12, gap=0)
    place(staged(["PA", "PB", "PC", "PD"]), 14, 42)                                    # S complete at slot 11 (+2), pw before slot 44
    place([("DL", g) for g in range(4)], 20, 24, gap=0)
    # dP complete at slot 23 (+2), gw before slot 56; the Q units of a quad reuse the scratch registers (xx, ff) of its P units
    place(staged(["QA", "QB", "QC", "QD"], "TW"), 26, 54, after=lambda name, q: ("PD", q) if name == "QA" else None)
    place([("XW", g) for g in range(4)], 46, 56, gap=0)                                # dq complete at slot 43 (+2)
    # dropout decisions of the NEXT tile: a lane computes eight of its key pair's sixteen hashes (quads 0, 1), HD exchanges and places them
    place([(st, q) for st in ("HA", "HB", "HC", "HD") for q in range(2)], 57, 75)

    lines = ["#define BWD4H_ITER()", "  do {"]
    for m in range(N):
        w = "; ".join(work[m])
        lines.append(f"    {mf[m]}; {w}; SB();" if w else f"    {mf[m]}; SB();")
    lines.append("  } while (0)")
    width = max(len(l) for l in lines) + 1
    print("// generated by tools/gen/attn_bwd4h_phase.py - the pinned body of one query tile of emu_attn_bwd4h_kernel (one MFMA + the work behind it)")
    print("\n".join(l.ljust(width) + "\\" for l in lines[:-1]) + "\n" + lines[-1])
    print("// modelled issue cycles of the other work per slot: max %d, mean %.1f, slots over %d: %d" % (max(load), sum(load) / float(N), CAP, sum(1 for x in load if x > CAP)), file=sys.stderr)
    print(" ".join(str(x) for x in load), file=sys.stderr)


if __name__ == "__main__":
    main()
