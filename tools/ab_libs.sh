#!/bin/bash
# A/B of library builds on one box: LIBS="name=path ..." (default: the in-tree build), WLS = bench_extra.py workload filters
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2; do
for spec in ${LIBS:-main=}; do
  name=${spec%%=*}; path=${spec#*=}
  if [ -n "$path" ]; then export SDRTERM_HIP_LIB=$PWD/$path; else unset SDRTERM_HIP_LIB; fi
  IFS='|' read -ra W <<< "${WLS:-dec=2|dec=4}"
  for wl in "${W[@]}"; do
    python tools/bench_extra.py "$wl" | python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$name', d['workload'], d['stage_a'], d['ms_per_step'], d['kernel_ms'])"
  done
done
done
