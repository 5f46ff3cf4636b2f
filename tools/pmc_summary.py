#!/usr/bin/env python3
"""Per-dispatch averages of the rocprofv3 counter passes written by tools/pmc.sh (chain kernels only)."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]
for pas in sorted(os.listdir(out)):
    d = os.path.join(out, pas)
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r.get('Kernel_Name', '')
            if 'chain' not in k:
                continue
            acc[k.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        for c, v in sorted(cs.items()):
            print(f'{pas:6s} {k:60s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}')
