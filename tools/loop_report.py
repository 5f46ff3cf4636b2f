#!/usr/bin/env python3
"""Steady-state loop of every chain_i8w_kernel instantiation in a gfx950 assembly file (hipcc -save-temps): the innermost
backward-branch range with the most matrix instructions; prints instructions, matrix instructions, scratch accesses (spill
traffic) and full `s_waitcnt vmcnt(0)` waits (each one drains the raw-stream prefetch ring) inside it.
usage: tools/loop_report.py file.s [name filter]"""
import re, sys

def functions(lines):
    name, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name, start = m.group(1), i
        elif l.startswith('.Lfunc_end') and name:
            yield name, lines[start:i]
            name = None

def report(name, body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\w+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\w+)', l) or re.match(r'\s+s_branch\s+(\.LBB\w+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    # innermost loops only (no other backward branch inside), the one with the most matrix instructions
    best = None
    for lo, hi in loops:
        if any((a, b) != (lo, hi) and lo <= a and b <= hi for a, b in loops):
            continue
        seg = body[lo:hi + 1]
        nm = sum('v_mfma' in x for x in seg)
        if best is None or nm > best[0]:
            best = (nm, len(seg), seg)
    if not best:
        return
    seg = best[2]
    ins = sum(bool(re.match(r'^\t[a-z]', x)) for x in seg)
    print(f'{name[:70]:70s} loop: {ins:5d} instr {best[0]:4d} mfma {sum("scratch_" in x for x in seg):3d} scratch '
          f'{sum("vmcnt(0)" in x for x in seg):2d} vmcnt(0)')

if __name__ == '__main__':
    lines = open(sys.argv[1]).read().split('\n')
    flt = sys.argv[2] if len(sys.argv) > 2 else 'chain_i8w'
    for name, body in functions(lines):
        if flt in name:
            report(name, body)
