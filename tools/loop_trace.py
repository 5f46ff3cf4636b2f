#!/usr/bin/env python3
"""Instruction-class trace of the largest loop body holding matrix instructions in an assembly file (after PROBE=1 tools/cc_i8w.sh):
M matrix, v vector ALU, d LDS read, D LDS write, g global load, G global store, s scalar, w s_waitcnt, b barrier, n s_nop --
how matrix and vector instructions alternate in one wave's instruction stream."""
import re, sys
f = sys.argv[1] if len(sys.argv) > 1 else '/tmp/asm/kernels_i8w-hip-amdgcn-amd-amdhsa-gfx950.s'
L = open(f).read().split('\n')
labels = {}
for i, l in enumerate(L):
    m = re.match(r'^(\.LBB\w+):', l)
    if m:
        labels[m.group(1)] = i
cands = []
for i, l in enumerate(L):
    m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\w+)', l) or re.match(r'\s+s_branch\s+(\.LBB\w+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = L[labels[m.group(1)]:i + 1]
        nm = sum('v_mfma' in x for x in seg)
        if nm:
            cands.append((sum('s_cbranch' in x for x in seg), -nm, labels[m.group(1)], i))
cands.sort()
_, nm, lo, hi = cands[int(sys.argv[2]) if len(sys.argv) > 2 else 0]
out = []
for l in L[lo:hi + 1]:
    m = re.match(r'^\t([a-z_0-9]+)', l)
    if not m:
        continue
    op = m.group(1)
    c = ('M' if op.startswith('v_mfma') else 'w' if op == 's_waitcnt' else 'b' if op == 's_barrier' else 'n' if op == 's_nop' else
         'd' if op.startswith('ds_read') else 'D' if op.startswith('ds_') else 'g' if op.startswith('global_load') else
         'G' if op.startswith('global_store') else 's' if op.startswith('s_') else 'v')
    out.append(c)
t = ''.join(out)
print(f'lines {lo}-{hi}: {len(t)} instructions, {-nm} matrix')
for i in range(0, len(t), 160):
    print(t[i:i + 160])
