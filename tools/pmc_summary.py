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

# traffic record for bench.py: chain kernel of the headline workload
import json, re
fetch = write = None
kname = None
for pas in ('fetch', 'write'):
    d = os.path.join(out, pas)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
                if 'chain' in r.get('Kernel_Name', '') and r['Counter_Name'] == ('FETCH_SIZE' if pas == 'fetch' else 'WRITE_SIZE')
                and float(r['Counter_Value']) > 1000]
        names = [r['Kernel_Name'] for r in csv.DictReader(open(f))
                 if 'chain' in r.get('Kernel_Name', '') and r['Counter_Name'] == ('FETCH_SIZE' if pas == 'fetch' else 'WRITE_SIZE')
                 and float(r['Counter_Value']) > 1000]
        if vals:
            # the workload's kernel is the one that moves the most bytes (sdrt_create also launches the fp64 kernel once)
            top = max(vals)
            sel = [v for v in vals if v > 0.5 * top]
            if pas == 'fetch':
                fetch = sum(sel) / len(sel)
                kname = names[vals.index(top)]
            else:
                write = sum(sel) / len(sel)
if fetch is not None and write is not None:
    m = re.search(r'(chain\w*kernel)', kname)
    rec = {'kernel': m.group(1) if m else kname, 'chunks_per_step_per_gpu': int(os.environ.get('BENCH_CHUNKS', '32768')),
           'fetch_size_kb': fetch, 'write_size_kb': write,
           'hbm_bytes_per_launch': int((2.0 * fetch + write) * 1024),
           'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md section HBM); WRITE_SIZE as is'}
    json.dump(rec, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1)
    print('traffic record:', rec)
