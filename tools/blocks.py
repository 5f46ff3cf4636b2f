#!/usr/bin/env python3
"""Basic blocks of one kernel in a -save-temps assembly file with their instruction mix (matrix, LDS, memory, scratch, waits):
   tools/blocks.py file.s <substring of the kernel's mangled name> [min instructions]"""
import re, sys
src = open(sys.argv[1]).read()
key = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 30
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", src, re.M)
i = m.start()
j = src.index('.end_amdhsa_kernel', i) if '.end_amdhsa_kernel' in src[i:] else len(src)
j = src.index('s_endpgm', i)
blocks = [['entry', []]]
for ln in src[i:j].split('\n'):
    s = ln.strip()
    if re.match(r'^\.LBB\d+_\d+:', s):
        blocks.append([s.split(':')[0], []])
    elif s and not s.startswith(';') and not s.startswith('.') and not s.endswith(':'):
        blocks[-1][1].append(s)
for name, ins in blocks:
    if len(ins) < minn:
        continue
    c = lambda p: sum(1 for x in ins if re.match(p, x))
    mf = c('v_mfma')
    print(f'{name:12s} n {len(ins):5d} mfma {mf:4d} ds_read {c("ds_read"):4d} ds_write {c("ds_write"):3d} vmem_ld {c("global_load|buffer_load"):3d} vmem_st {c("global_store|buffer_store"):3d} '
          f'scratch {c("scratch_"):3d} f64 {sum(1 for x in ins if "_f64" in x):4d} accvgpr {c("v_accvgpr"):4d} waitcnt {c("s_waitcnt"):3d} valu {c("v_") - mf:4d} nop {c("s_nop"):3d}')
