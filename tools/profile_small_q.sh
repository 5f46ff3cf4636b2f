#!/bin/bash
# The small-decimation part of a round's measurement set (chain_sq_kernel): bench lines, kernel statistics under rocprofv3, counter
# records of fm2 / fm4 / fm8 / fm16 / fm16b (uint8) -> gpurun_out/prof_<tag>_smallq/ (copy what is to be judged into profiles/).
# usage: tools/profile_small_q.sh r05
tag=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_${tag}_smallq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in fm2 fm4 fm8 fm16 fm16b fm2b; do
    python3 "$ROOT/bench.py" --workload $w --steps 300 > "$OUT/${w}_n1.json" 2> "$OUT/${w}.err" || { echo "$w failed"; exit 1; }
    echo "$w line done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --workload fm2 --steps 300 --warmup 20 --no-cpu-baseline --no-same-arithmetic > "$OUT/fm2_under_rocprof.json" 2> "$OUT/rocprof.err" || exit 1
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats_fm2.csv"
rm -rf "$OUT/stats"
echo "rocprof stats done"
for w in fm2 fm16 fm8 fm4 fm16b; do
    BENCH_ARGS="--workload $w" bash "$ROOT/tools/pmc.sh" > "$OUT/pmc_$w.log" 2>&1
    echo "pmc $w done"
done
cp "$ROOT"/gpurun_out/pmc/summary_fm{2,4,8,16,16b}.txt "$ROOT"/gpurun_out/pmc/pmc_record_fm{2,4,8,16,16b}.json "$OUT"/ 2>/dev/null
echo "all done"
