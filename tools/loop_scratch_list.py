#!/usr/bin/env python3
"""Scratch accesses inside the steady-state loop (smallest backward-branch range holding matrix instructions) of the first
chain_i8w_kernel instantiation in an assembly file, with their offsets from the loop header.  After tools/cc_i8w.sh PROBE=1."""
import re, sys
f = sys.argv[1] if len(sys.argv) > 1 else '/tmp/asm/kernels_i8w-hip-amdgcn-amd-amdhsa-gfx950.s'
L = open(f).read().split('\n')
labels = {}
for i, l in enumerate(L):
    m = re.match(r'^(\.LBB\w+):', l)
    if m:
        labels[m.group(1)] = i
best = None
for i, l in enumerate(L):
    m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\w+)', l) or re.match(r'\s+s_branch\s+(\.LBB\w+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        lo = labels[m.group(1)]
        seg = L[lo:i + 1]
        nm = sum('v_mfma' in x for x in seg)
        if nm and (best is None or len(seg) < best[2]):
            best = (lo, i, len(seg), nm)
lo, hi, _, nm = best
seg = L[lo:hi + 1]
print(f'loop at lines {lo}-{hi}: {sum(bool(re.match(chr(9) + "[a-z]", x)) for x in seg)} instructions, {nm} matrix instructions, '
      f'{sum("scratch_" in x for x in seg)} scratch accesses, {sum("vmcnt(0)" in x for x in seg)} vmcnt(0), {sum("s_barrier" in x for x in seg)} barriers')
for i in range(lo, hi + 1):
    if 'scratch_' in L[i]:
        print(f'  +{i - lo}: {L[i].strip()[:100]}')
