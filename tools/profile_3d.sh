#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace + stats and FETCH_SIZE / WRITE_SIZE passes of tools/bench_3d.py (245 760-cell 3-D
# mesh, records beyond L2) -> gpurun_out/<label>_3d_summary.json.  Traffic here is EXPECTED above the algorithmic
# 56 B per particle-step: every cell visit of a wave may fetch a 256-byte record from the Infinity Cache / HBM.
set -u
LABEL="${1:-r02}"
OUT="gpurun_out/${LABEL}_3d"
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/bench_3d.py > "$OUT/bench_under_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/bench_3d.py > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 tools/bench_3d.py > "$OUT/pmc_write.log" 2>&1
python3 tools/prof_summary.py "${LABEL}_3d" "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" > "gpurun_out/${LABEL}_3d_summary.json" 2> "$OUT/summary.err"
grep kernel_ms "$OUT/bench_under_trace.log"
head -c 2500 "gpurun_out/${LABEL}_3d_summary.json"
find "$OUT" -name "*.csv" -size +2M -delete
