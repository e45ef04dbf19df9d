#!/usr/bin/env python3
"""Emits the two pinned half-iterations of emu_attn_fwd2_kernel (hoisdf_amd/csrc/attention_emu.hip):
  FWD2_S(KB, pA, pB)   24 MFMAs of S(t + 1) = K(t + 1) . Q^T (four 16-wide d steps x six plane products) with the softmax / dropout /
                       three-way split of tile t (8 score pairs x 4 units) pinned behind them;
  FWD2_PV(VB, ...)     24 MFMAs of O^T += V(t)^T . P(t)^T (two key half-tiles x two 32-wide d blocks x six products) with the running
                       maximum of tile t + 1, its dropout decisions (8 hashes, DROP only), the staging writes of K(t + 2) / V(t + 1)
                       and the loads of K(t + 3) / V(t + 2) behind them.
`python tools/gen/attn_fwd2_phase.py` prints attn_fwd2_phase.inc."""
PROD = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]          # (x plane, y plane): small terms first


def emit(name, params, slots):
    lines = [f"#define {name}({params})", "  do {"]
    for m, w in slots:
        lines.append(f"    {m}; {w}; SB();" if w else f"    {m}; SB();")
    lines.append("  } while (0)")
    width = max(len(l) for l in lines) + 1
    return "\n".join(l.ljust(width) + "\\" for l in lines[:-1]) + "\n" + lines[-1]


def phase_s():
    # MFMA m = 6 j + k: s_nxt += K_x(j) . Q_y(j); fragments of step j live in set (j & 1); step j + 1 is read during step j
    work = {}
    add = lambda s, w: work.setdefault(s, []).append(w)
    for j in range(1, 4):
        for p in range(3):
            add(6 * (j - 1) + 1 + p, f"kf[{j & 1}][{p}] = KFRAG(KB, {p}, {j})")
    units = []
    for pr in range(8):
        units += [f"PE1({pr})", f"PE2({pr})", f"PE3({pr})", f"PE4({pr})"]
    for i, u in enumerate(units):               # 32 units over 24 slots
        add(i * 24 // 32, u)
    slots = []
    for j in range(4):
        for k, (x, y) in enumerate(PROD):
            m = 6 * j + k
            slots.append((f"s_nxt = MB(kf[{j & 1}][{x}], qf[{j}][{y}], s_nxt)", "; ".join(work.get(m, []))))
    return emit("FWD2_S", "KB", slots)


def phase_pv():
    # MFMA m = 12 jj + 6 dt + k: o[dt] += V_x(jj, dt) . P_y(jj); V fragments of group g = 2 jj + dt in set (g & 1), next group read ahead
    work = {}
    add = lambda s, w: work.setdefault(s, []).append(w)
    for g in range(1, 4):
        jj, dt = g >> 1, g & 1
        for p in range(3):
            add(6 * (g - 1) + 1 + p, f"vf[{g & 1}][{p}] = VFRAG(VB, {p}, {dt}, {jj})")
    for i in range(4): add(4 + i, f"PM({i})")        # (asm reads of the S accumulator: kept >= 4 MFMAs behind the last S MFMA)
    units = []
    for pr in range(8):
        units += [f"PH1({pr})", f"PH2({pr})", f"PH3({pr})"]
    units += [f"STK({p})" for p in range(3)] + [f"STV({p})" for p in range(3)] + [f"LDK({p})" for p in range(3)] + [f"LDV({p})" for p in range(3)]
    n = len(units)
    for i, u in enumerate(units):
        add(i * 24 // n, u)
    slots = []
    for g in range(4):
        jj, dt = g >> 1, g & 1
        for k, (x, y) in enumerate(PROD):
            m = 6 * g + k
            slots.append((f"o[{dt}] = MB(vf[{g & 1}][{x}], pw[{jj}][{y}], o[{dt}])", "; ".join(work.get(m, []))))
    return emit("FWD2_PV", "VB", slots)


PROD2 = [(1, 0), (0, 1), (0, 0)]                               # two planes per operand (bf16 hi + lo): three products


def phase_s2():
    """FWD2_S2: the 12 MFMAs of S(t + 1) over two planes with the softmax / two-way split of tile t (8 pairs x 3 units)"""
    work = {}
    add = lambda s, w: work.setdefault(s, []).append(w)
    for j in range(1, 4):
        for p in range(2):
            add(3 * (j - 1) + 1 + p, f"kf[{j & 1}][{p}] = KFRAG(KB, {p}, {j})")
    units = []
    for pr in range(8):
        units += [f"PE1({pr})", f"PE2({pr})", f"PE3B({pr})"]
    for i, u in enumerate(units):
        add(i * 12 // len(units), u)
    slots = []
    for j in range(4):
        for k, (x, y) in enumerate(PROD2):
            m = 3 * j + k
            slots.append((f"s_nxt = MB(kf[{j & 1}][{x}], qf[{j}][{y}], s_nxt)", "; ".join(work.get(m, []))))
    return emit("FWD2_S2", "KB", slots)


def phase_pv2():
    work = {}
    add = lambda s, w: work.setdefault(s, []).append(w)
    for g in range(1, 4):
        jj, dt = g >> 1, g & 1
        for p in range(2):
            add(3 * (g - 1) + 1 + p, f"vf[{g & 1}][{p}] = VFRAG(VB, {p}, {dt}, {jj})")
    for i in range(4): add(3 + i, f"PM({i})")        # (asm reads of the S accumulator: kept >= 3 MFMAs behind the last S MFMA)
    stage = [f"STK({p})" for p in range(2)] + [f"STV({p})" for p in range(2)] + [f"LDK({p})" for p in range(2)] + [f"LDV({p})" for p in range(2)]
    units = []
    for pr in range(8):                          # the dropout decisions of tile t + 1 (DROP only: empty macros otherwise), the staging spread between them
        units += [f"PH1({pr})", f"PH2({pr})", f"PH3({pr})", stage[pr]]
    for i, u in enumerate(units):
        add(i * 12 // len(units), u)
    slots = []
    for g in range(4):
        jj, dt = g >> 1, g & 1
        for k, (x, y) in enumerate(PROD2):
            m = 3 * g + k
            slots.append((f"o[{dt}] = MB(vf[{g & 1}][{x}], pw[{jj}][{y}], o[{dt}])", "; ".join(work.get(m, []))))
    return emit("FWD2_PV2", "VB", slots)


if __name__ == "__main__":
    print("// generated by tools/gen/attn_fwd2_phase.py - the pinned half-iterations of emu_attn_fwd2_kernel (one MFMA + the units behind it)")
    print(phase_s())
    print(phase_pv())
    print("// two planes per operand (bf16 hi + lo, three products per product): the 16-bit-operand evaluation kernel of BASELINE configs[4]")
    print(phase_s2())
    print(phase_pv2())
