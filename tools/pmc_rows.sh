#!/bin/bash
# FETCH_SIZE / clock of the 8-row and simo-like workloads
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
for w in "configs[2]-like" "single VFO FM dec=64 uint8 --correct-iq"; do
  tag=$(echo "$w" | tr -c 'a-zA-Z0-9' '_')
  rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d /tmp/p8_$tag -- python3 $ROOT/tools/bench_extra.py "$w" > /tmp/p8_$tag.log 2>&1
  f=$(find /tmp/p8_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$w" <<'PY'
import csv, sys, collections
f, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
print('==', w)
for k, d in acc.items():
    if 'chain' in k or 'post' in k or 'iq_' in k:
        # sum per dispatch over the 8 XCD rows is done by rocprof? print mean of per-dispatch sums
        print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
done
