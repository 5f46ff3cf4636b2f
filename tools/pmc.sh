#!/bin/bash
# Hardware-counter passes over bench.py (run on the GPU box).  One rocprofv3 invocation per counter group, counters
# only with --kernel-trace, program = python3 directly (no wrappers), as the pool requires.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=/tmp/pmc_$$                 # raw profiler output stays on the box (it exceeds what gpurun copies back)
KEEP=$ROOT/gpurun_out/pmc
mkdir -p "$OUT" "$KEEP"
cd /tmp
# BENCH_ARGS="--workload simo8" tools/pmc.sh  -> gpurun_out/pmc/pmc_record_simo8.json (copy to profiles/r04_pmc_record_simo8.json)
ARGS="--steps 3 --warmup 1 --settle-seconds 0.5 --long-seconds 0 --no-cpu-baseline --no-same-arithmetic --no-parity ${BENCH_ARGS:-}"
WL=$(echo " ${BENCH_ARGS:-}" | sed -n 's/.*--workload[ =]\([a-z0-9]*\).*/\1/p'); WL=${WL:-fm64}
case " ${BENCH_ARGS:-} " in *" --correct-iq "*) case $WL in simo8) ;; *) WL=${WL}_iq ;; esac ;; esac
case " ${BENCH_ARGS:-} " in *" --faithful "*) WL=${WL}_faithful ;; esac
export BENCH_WORKLOAD=$WL
pass() {
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
    echo "$name exit=$?"
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT
pass sq3 SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_WR
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE GRBM_GUI_ACTIVE
python3 "$ROOT/tools/pmc_summary.py" "$OUT" | tee "$OUT/summary_$WL.txt"
cp "$OUT"/pmc_record.json "$OUT/pmc_record_$WL.json" 2>/dev/null
cp "$OUT"/summary_$WL.txt "$OUT"/pmc_record_$WL.json "$KEEP"/ 2>/dev/null
for f in "$OUT"/*.log; do cp "$f" "$KEEP/$(basename "$f" .log)_$WL.log"; done
rm -rf "$OUT"
