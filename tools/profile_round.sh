#!/bin/bash
# The round's measurement set on one GPU box -> gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/):
#   bench line, bench line under rocprofv3 with the kernel-trace statistics, counter passes + record, phase stamps
#   (diagnostic build libsdrterm_hip_exp.so made with EXTRA=-DSDRT_STAMPS), determinism.
# usage: tools/profile_round.sh r03
tag=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$tag
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 300 --warmup 20 --no-cpu-baseline --no-same-arithmetic > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err" || exit 1
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats"
echo "rocprof stats done"
bash "$ROOT/tools/pmc.sh" > "$OUT/pmc.log" 2>&1
cp "$ROOT/gpurun_out/pmc/summary_fm64.txt" "$OUT/pmc_summary_fm64.txt" 2>/dev/null
cp "$ROOT/gpurun_out/pmc/pmc_record_fm64.json" "$OUT/pmc_record_fm64.json" 2>/dev/null
echo "pmc done"
if [ -f "$ROOT/sdrterm_amd/libsdrterm_hip_exp.so" ]; then
    SDRTERM_HIP_LIB=$ROOT/sdrterm_amd/libsdrterm_hip_exp.so python3 "$ROOT/tools/stamps_wide.py" 131072 15000 450 > "$OUT/i8w_phase_stamps_after_2s_load.txt" 2>&1
    echo "stamps done"
fi
python3 "$ROOT/tools/determinism.py" > "$OUT/determinism.txt" 2>&1
tail -2 "$OUT/determinism.txt"
# the other bench workloads (one line each) and the secondary workloads of tools/bench_extra.py
for w in fm64f fm64b fm16 fm2 simo8 simo64 f32x256; do
    python3 "$ROOT/bench.py" --workload $w --steps 300 > "$OUT/${w}_n1.json" 2> "$OUT/${w}.err" || echo "$w failed"
done
python3 "$ROOT/bench.py" --correct-iq --steps 300 > "$OUT/fm64_correct_iq.json" 2> "$OUT/fm64_correct_iq.err" || echo "fm64 --correct-iq failed"
# float32 records in the reference's own complex64 arithmetic (sdrt_config.faithful_f32)
python3 "$ROOT/bench.py" --workload fm64f --faithful --steps 20 --warmup 2 > "$OUT/fm64f_faithful_n1.json" 2> "$OUT/fm64f_faithful.err" || echo "fm64f --faithful failed"
python3 "$ROOT/bench.py" --workload f32x256 --faithful --steps 20 --warmup 2 > "$OUT/f32x256_faithful_n1.json" 2> "$OUT/f32x256_faithful.err" || echo "f32x256 --faithful failed"
# the second look of stage_a = auto costs the headline nothing: the same run without it
SDRTERM_DIAG=refine=0 python3 "$ROOT/bench.py" --steps 300 --no-cpu-baseline --no-same-arithmetic > "$OUT/fm64_without_second_look.json" 2> /dev/null || echo "refine=0 failed"
# counter records of the other workloads (tools/pmc.sh), each stamped with the commit
for a in "--workload fm2" "--workload fm16" "--workload fm64b" "--workload simo8" "--workload simo64" "--workload f32x256" "--workload f32x256 --faithful" "--workload fm64f --faithful"; do
    BENCH_ARGS="$a" bash "$ROOT/tools/pmc.sh" > /dev/null 2>&1
done
cp "$ROOT"/gpurun_out/pmc/summary_*.txt "$ROOT"/gpurun_out/pmc/pmc_record_*.json "$OUT"/ 2>/dev/null
python3 "$ROOT/tools/bench_extra.py" > "$OUT/secondary_workloads.jsonl" 2> "$OUT/secondary.err"
echo "workloads done"
