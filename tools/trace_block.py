#!/usr/bin/env python3
"""One basic block of a kernel as a string of instruction classes (M matrix, v vector, a accvgpr move, d / D LDS read / write,
g / G global load / store, w waitcnt, n nop, s scalar):  tools/trace_block.py file.s <kernel name substring> <.LBBx_y>"""
import re, sys, textwrap
src = open(sys.argv[1]).read()
m = re.search(r"^(\S*" + re.escape(sys.argv[2]) + r"\S*):", src, re.M)
body = src[m.start():src.index('s_endpgm', m.start())].split('\n')
lab = sys.argv[3] + ':'
i = next(k for k, l in enumerate(body) if l.strip().startswith(lab))
out = []
for l in body[i + 1:]:
    l = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', l):
        break
    if not l or l.startswith(';') or l.startswith('.'):
        continue
    for pat, ch in (('v_mfma', 'M'), ('v_accvgpr', 'a'), ('v_', 'v'), ('ds_read', 'd'), ('ds_', 'D'), ('global_load', 'g'), ('global_store', 'G'), ('scratch_', 'S'),
                    ('s_waitcnt', 'w'), ('s_nop', 'n'), ('s_', 's')):
        if l.startswith(pat):
            out.append(ch)
            break
    else:
        out.append('?')
print('\n'.join(textwrap.wrap(''.join(out), 160)))
