#!/bin/bash
# Wave-priority assignments of chain_i8w_kernel (kernels_i8w.hip: SDRT_SCHED_EXP) through bench.py on the diagnostic build:
#   make -C sdrterm_amd/csrc OUT=../libsdrterm_hip_exp.so OBJ=_obj_exp EXTRA=-DSDRT_SCHED_EXP
# One bench line per assignment -> gpurun_out/sched_exp.jsonl; summary on stdout.  GPU box.
cd "$(dirname "$0")/.." || exit 1
out=${1:-gpurun_out/sched_exp.jsonl}
: > "$out"
for k in ${SCHEDS:-0 1 2 3 4 5 6 0}; do
    SDRTERM_HIP_LIB=$PWD/sdrterm_amd/libsdrterm_hip_exp.so SDRTERM_DIAG=sched=$k python bench.py --steps 20 --warmup 3 --settle-seconds 1.5 --long-seconds 0 --no-cpu-baseline \
        | python -c "import sys, json; d = json.loads(sys.stdin.readline()); d['sched'] = $k; print(json.dumps(d))" >> "$out" || exit 1
    python - "$out" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"sched {d['sched']}: ms_per_step {d['ms_per_step']:.4f}  chain {d['kernels_ms']['chain_kernel']:.4f}  post {d['kernels_ms']['post_kernel']:.4f}  path_frac {d['roofline']['path_frac']:.4f}", flush=True)
PY
done
