#!/usr/bin/env python3
"""Emits the PHASE macro of emu_kc2_kernel (hoisdf_amd/csrc/gemm_emu.hip) from a slot table: 48 MFMAs of a phase, each followed
by the staging / fragment-read units pinned behind it.  `python tools/gen/kc2_phase.py <variant>` prints the macro; the shipped
variant is pasted into the kernel (the table is easier to audit than 48 hand-written lines)."""
import sys

# MFMA order of a phase: (A fragments, B fragments) x (i, j)
GROUPS = [("aX", "b2"), ("aX", "b1"), ("aX", "bC"), ("aY", "bN"), ("aX", "b1"), ("aX", "bN")]
NOTE = {0: "x0 y2 of slab s - 1", 8: "x0 y1", 16: "x0 y0", 24: "x2 y0 of slab s", 32: "x1 y1", 40: "x1 y0"}


def reads():
    t = {}
    for i in range(4): t.setdefault(i, []).append(f"aY[{i}] = LDA(cur, 2, {i})")
    for j in range(2): t.setdefault(4 + j, []).append(f"bN[{j}] = LDB(cur, 0, {j})")
    for j in range(2): t.setdefault(8 + j, []).append(f"b2[{j}] = LDB(cur, 2, {j})")
    for j in range(2): t.setdefault(16 + j, []).append(f"b1[{j}] = LDB(cur, 1, {j})")
    for i in range(4): t.setdefault(24 + i, []).append(f"aX[{i}] = LDA(cur, 1, {i})")
    for i in range(4): t.setdefault(32 + i, []).append(f"aY[{i}] = LDA(cur, 0, {i})")
    return t


def add(t, slot, w):
    t.setdefault(slot, []).append(w)


def variant(v):
    """slot -> units.  Fixed: the 18 fragment reads (reads()), the three weight-image writes / loads.  Per item i (a quad of one
    of the thread's four rows): UI + 8 conversion units (two pairs x U1..U4), three 8-byte LDS writes, the reload."""
    t = reads()
    for q, sl in ((0, 6), (1, 7), (2, 10)): add(t, sl, f"STB(nxt, {q})")
    for q in range(3): add(t, (14 + 8 * q) if v == 3 else (11 + q), f"LDGB({q}, (s) + 2)")   # 3: the image loads spread between the activation loads
    if v == 1 or v == 3:      # spread: one conversion unit per MFMA from slot 2; an item's writes and reload right behind its last unit
        start, step = 2, 8
    elif v == 2:    # late: the conversion starts at slot 12 (loads have had 10 more MFMAs to land)
        start, step = 12, 8
    else:
        raise SystemExit("variant 1..3")
    for i in range(4):
        s0 = start + step * i
        units = [f"UI({i}, (s) + 1); U1({2 * i})", f"U2({2 * i})", f"U3({2 * i})", f"U4({2 * i})",
                 f"U1({2 * i + 1})", f"U2({2 * i + 1})", f"U3({2 * i + 1})", f"U4({2 * i + 1})"]
        for k, w in enumerate(units): add(t, s0 + k, w)
        e = s0 + 8
        add(t, e, f"STA(nxt, {i}, t0, 0)"); add(t, e + 1, f"STA(nxt, {i}, t1, 1)"); add(t, e + 2, f"STA(nxt, {i}, t2, 2)")
        add(t, e, f"LDGA({i}, (s) + 2)")
    assert max(t) < 48
    return t


def emit(v):
    t = variant(v)
    lines = ["#define PHASE(cur, nxt, s, aX, aY, bC, bN)", "  do {"]
    for m in range(48):
        ax, bx = GROUPS[m // 8]
        i, j = (m % 8) // 2, m % 2
        w = "; ".join(t.get(m, [])) or "NOP_"
        l = f"    M1({ax}, {bx}, {i}, {j}, {w});"
        if m in NOTE: l += f"   /* {NOTE[m]} */"
        lines.append(l)
    lines.append("  } while (0)")
    width = max(len(l) for l in lines) + 1
    return "\n".join(l.ljust(width) + "\\" for l in lines[:-1]) + "\n" + lines[-1]


if __name__ == "__main__":
    print(emit(int(sys.argv[1]) if len(sys.argv) > 1 else 2))
