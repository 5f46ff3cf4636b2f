#!/usr/bin/env python3
"""Per-dispatch averages of the rocprofv3 counter passes written by tools/pmc.sh, and the record bench.py reads
(profiles/pmc_record.json): HBM bytes per launch of the dominant kernel (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
for gfx950, plus WRITE_SIZE), matrix-pipe busy fraction and the clock the chip held."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))        # kernel -> counter -> values
dur = defaultdict(list)
for pas in sorted(os.listdir(out)):
    d = os.path.join(out, pas)
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get('Kernel_Name', '')
            if 'chain' not in k and 'post' not in k and 'iq_' not in k and 'preslice' not in k and 'f32c' not in k:
                continue
            acc[k.replace('(anonymous namespace)::', '').split('(')[0][:64]][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get('Kernel_Name', '')
            if 'chain' in k or 'post' in k or 'iq_' in k or 'preslice' in k or 'f32c' in k:
                dur[k.replace('(anonymous namespace)::', '').split('(')[0][:64]].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6)

def big(vals):
    """the workload's launches (sdrt_create also runs small one-chunk launches)"""
    top = max(vals)
    sel = [v for v in vals if v >= 0.5 * top]
    return sum(sel) / len(sel)

rec = {}
for k, cs in sorted(acc.items()):
    for c, v in sorted(cs.items()):
        print(f'{k:64s} {c:28s} n={len(v):3d} mean(workload launches)={big(v):.6g}')
    if k in dur:
        print(f'{k:64s} {"duration_ms (profiled)":28s} n={len(dur[k]):3d} mean(workload launches)={big(dur[k]):.6g}')
    name = re.search(r'(chain\w*kernel|post\w*kernel|iq_\w*kernel|f32_preslice_kernel|f32c_\w*kernel)', k)
    if not name or 'FETCH_SIZE' not in cs or 'WRITE_SIZE' not in cs:
        continue
    e = {'kernel': name.group(1), 'fetch_size_kb': big(cs['FETCH_SIZE']), 'write_size_kb': big(cs['WRITE_SIZE'])}
    e['hbm_bytes_per_launch'] = int((2.0 * e['fetch_size_kb'] + e['write_size_kb']) * 1024)
    if 'GRBM_GUI_ACTIVE' in cs and k in dur:
        cyc = big(cs['GRBM_GUI_ACTIVE']) / 8.0                       # summed over the 8 XCDs
        e['clock_mhz'] = round(cyc / (big(dur[k]) * 1e-3) / 1e6, 1)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs:
            e['mfma_busy'] = round(big(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / (1024.0 * cyc), 4)   # 1024 SIMDs
        if 'SQ_VALU_MFMA_COEXEC_CYCLES' in cs and 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and big(cs['SQ_VALU_MFMA_BUSY_CYCLES']) > 0:
            e['mfma_valu_coexec'] = round(big(cs['SQ_VALU_MFMA_COEXEC_CYCLES']) / big(cs['SQ_VALU_MFMA_BUSY_CYCLES']), 4)
    rec[e['kernel']] = e
chain = next((v for k, v in rec.items() if k.startswith('chain')), None) or rec.get('f32c_fused_kernel') or rec.get('f32c_seq_kernel')
if chain:
    defaults = {'fm64': 131072, 'fm64_iq': 131072, 'fm64f': 131072, 'fm64b': 65536, 'fm64b_iq': 65536, 'fm16': 16384, 'fm2': 2048, 'simo8': 4096, 'simo64': 512, 'f32x256': 128, 'fm64f_faithful': 131072, 'f32x256_faithful': 1024}
    wl = os.environ.get('BENCH_WORKLOAD', 'fm64')
    # which build the counters belong to: tools/gpu.sh leaves the commit (and whether the tree was clean) in .build_commit before the
    # snapshot travels to the GPU box, where there is no .git
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        stamp = open(os.path.join(root, '.build_commit')).read().split()
        commit, clean = stamp[0], (stamp[1] == 'clean' if len(stamp) > 1 else None)
    except (OSError, IndexError):
        commit, clean = None, None
    top = dict(chain, workload=wl, commit=commit, tree_clean=clean, chunks_per_step_per_gpu=int(os.environ.get('BENCH_CHUNKS', defaults.get(wl, 131072))), other_kernels={k: v for k, v in rec.items() if v is not chain},
               note='rocprofv3 --pmc, one pass per counter group with --kernel-trace; FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced '
                    'reads, MI355X_MICROARCH.md section HBM); WRITE_SIZE as is; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x '
                    'GRBM_GUI_ACTIVE / 8); clock = GRBM_GUI_ACTIVE / 8 / profiled duration')
    json.dump(top, open(os.path.join(out, 'pmc_record.json'), 'w'), indent=1)
    print('record:', json.dumps(top))
