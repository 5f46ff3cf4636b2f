#!/bin/bash
# A/B of two library builds on one box: old = sdrterm_amd/libsdrterm_hip_old.so
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export SDRTERM_HIP_LIB=$PWD/sdrterm_amd/libsdrterm_hip_old.so; else unset SDRTERM_HIP_LIB; fi
  for wl in "dec=32 uint8" "configs[2]-like"; do
    python tools/bench_extra.py "$wl" | python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$lib', d['workload'], d['ms_per_step'], d['kernel_ms'])"
  done
  for w in simo64 simo8; do python bench.py --workload $w --steps 100 --warmup 3 --settle-seconds 1.0 --long-seconds 0 --no-cpu-baseline --no-same-arithmetic --no-parity | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib $w', d['ms_per_step'], d['kernels_ms'])"; done
done
done
