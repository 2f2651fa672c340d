#!/usr/bin/env bash
# GPU box: ONE profiling recipe for any of the tools (replaces the per-case profile_*.sh / pmc_*.sh scripts of rounds 1-3):
#   tools/pmc.sh LABEL [--sq] -- python3 <script> <args ...>
# pass 1: rocprofv3 --kernel-trace --stats; passes 2, 3: --pmc FETCH_SIZE / --pmc WRITE_SIZE, each in its own run and never
# combined with sys / hip / hsa tracing (MI355X_MICROARCH.md, and gpurun refuses the combination); --sq: four more passes with
# the SQ instruction / wait counters (sets of <= 8: larger ones abort rocprofv3 with "exceeds the capabilities of the hardware").
# The program after "--" is python3 itself -- never env / bash -c / a launcher that re-execs (the profiler's preloaded library
# has initialised the GPU before the program starts).  Summary: gpurun_out/LABEL_summary.json (tools/prof_summary.py: per step
# kernel calls, average duration, HBM bytes per launch with the gfx950 corrections of profiles/r03_pmc_hbm.json: FETCH_SIZE KB x
# 1024 x 1.9975, WRITE_SIZE KB x 1024 x 0.9934, calibrated on the zero-cycle launch: tools/calib.py), SQ counters per launch
# and per 64-particle tile: gpurun_out/LABEL_sq.txt.  Copy what is to be judged into profiles/.
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
LABEL="$1"; shift
SQ=0; [ "${1:-}" = "--sq" ] && { SQ=1; shift; }
[ "${1:-}" = "--" ] && shift
OUT="gpurun_out/$LABEL"; rm -rf "$OUT"; mkdir -p "$OUT"
run() { d=$1; shift; timeout -s KILL 600 rocprofv3 "$@" --output-format csv -d "$OUT/$d" -- "${CMD[@]}" > "$OUT/$d.log" 2>&1 < /dev/null || echo "$d failed: $(tail -2 "$OUT/$d.log" | cut -c1-200)"; }
CMD=("$@")
run stats --kernel-trace --stats
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
python3 tools/prof_summary.py "$LABEL" "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" > "gpurun_out/${LABEL}_summary.json" 2> "$OUT/summary.err"
grep '^{' "$OUT/stats.log" | cut -c1-400
if [ $SQ = 1 ]; then
  SETS=(
   "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
   "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
   "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"
   "GRBM_GUI_ACTIVE GRBM_COUNT"
  )
  k=0
  for S in "${SETS[@]}"; do run "sq$k" --kernel-trace --pmc $S; k=$((k+1)); done
  python3 tools/prof_summary.py --sq "$OUT" > "gpurun_out/${LABEL}_sq.txt"
fi
python3 - "$LABEL" <<'PY'
import json, sys
d = json.load(open("gpurun_out/%s_summary.json" % sys.argv[1]))
for k in d.get("step_kernels", []):
    print(k["kernel"][:70], "calls", k["calls"], "avg_us", round(k["avg_us"], 2), "hbm_MB", round((k.get("hbm_bytes_per_launch") or 0) / 1e6, 1))
PY
find "$OUT" -name "*.csv" -size +4M -delete
