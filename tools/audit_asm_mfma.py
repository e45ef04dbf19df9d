#!/usr/bin/env python3
"""Hazard audit of a kernel whose MFMAs are inline asm (hipcc pads nothing around them): for every MFMA in the given .s range
  (a) no VALU / accvgpr instruction among the previous 2 instructions writes one of its A / B / C operand registers;
  (b) every non-MFMA reader (or writer) of an MFMA's VGPR destination sits at least 2 MFMAs behind the LAST MFMA that wrote it.
usage: audit_asm_mfma.py file.s kernel_symbol_substring"""
import re, sys


def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'^([va])\[(\d+):(\d+)\]$', tok)
    if m: return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r'^([va])(\d+)$', tok)
    if m: return {(m.group(1), int(m.group(2)))}
    return set()


def main():
    text = open(sys.argv[1]).read()
    pat = sys.argv[2]
    bad = 0
    for km in re.finditer(r'^(_ZN\S*' + pat + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', text, re.S | re.M):
        ins = []
        for l in km.group(2).split('\n'):
            t = l.strip()
            if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
            t = t.split(';')[0].strip()
            op, _, rest = t.partition(' ')
            ops = [x for x in re.split(r',\s*', rest.strip()) if x]
            ins.append((op, ops))
        last_writer = {}          # reg -> index (in MFMA count) of the last MFMA writing it
        nm = 0
        for i, (op, ops) in enumerate(ins):
            if 'mfma' in op:
                srcs = set().union(*[regs(o) for o in ops[1:4]])
                for back in (1, 2):
                    if i - back < 0: continue
                    pop, pops = ins[i - back]
                    if pop.startswith('v_') and 'mfma' not in pop and pops:
                        w = regs(pops[0])
                        if pop.startswith('v_pk') or pop.startswith('v_cvt_pk') or True:
                            if w & srcs:
                                print(f'  (a) {km.group(1)[:40]}: {pop} {pops[0]} writes an operand {back} instruction(s) ahead of MFMA #{nm}')
                                bad += 1
                nm += 1
                for r in regs(ops[0]):
                    if r[0] == 'v': last_writer[r] = nm
            else:
                touched = set().union(*[regs(o) for o in ops]) if ops else set()
                for r in touched:
                    if r in last_writer and nm - last_writer[r] < 2:
                        print(f'  (b) {km.group(1)[:40]}: {op} {" ".join(ops)[:60]} touches {r} {nm - last_writer[r]} MFMA(s) behind its producer (MFMA #{last_writer[r]})')
                        bad += 1
                        break
        print(km.group(1)[:60], 'MFMAs', nm, 'findings', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
