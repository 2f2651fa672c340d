#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace + stats of the bench command, then FETCH_SIZE / WRITE_SIZE in separate
# --pmc passes (never combined with sys/hip/hsa tracing), plus the same two counters on the zero-cycle
# calibration launch.  Everything lands under gpurun_out/<label>/; the condensed summary is
# gpurun_out/<label>_summary.json (copy it to profiles/).
set -u
LABEL="${1:-r01}"
OUT="gpurun_out/$LABEL"
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
BENCH="bench.py --no-cpu-baseline --steady-steps 0 --brownian-extra 0 --fused-extra 0 --anchor-particles 0"   # the bench command, timed region only (100 timed steps after 10 warm-up): the extras of the default run would mix other launches of the same kernel into the averages
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $BENCH > "$OUT/bench_under_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/cal_fetch" -- python3 tools/calib.py 1e7 10 > "$OUT/cal_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/cal_write" -- python3 tools/calib.py 1e7 10 > "$OUT/cal_write.log" 2>&1
python3 tools/prof_summary.py "$LABEL" "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/cal_fetch" "$OUT/cal_write" > "gpurun_out/${LABEL}_summary.json" 2> "$OUT/summary.err"
tail -3 "$OUT/bench_under_trace.log"
head -c 3000 "gpurun_out/${LABEL}_summary.json"
# raw CSVs are big: keep only the stats files and the summary for the merge back (<= 64 MiB)
find "$OUT" -name "*.csv" -size +2M -delete
