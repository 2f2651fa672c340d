#!/usr/bin/env bash
# GPU box: SQ counter passes (TCC/TCP sets of more than a few counters abort rocprofv3 with 'exceeds the capabilities
# of the hardware' and the aborted process then hangs: keep each set small and run the script under timeout) over tools/bench_3d.py (or PMC_CMD); prints per-launch means of the
# statistics-off step kernel, also per 64-particle tile.  usage: tools/pmc_3d.sh [label]
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
CMD="${PMC_CMD:-tools/bench_3d.py}"
OUT=gpurun_out/pmc_3d_${1:-x}; rm -rf "$OUT"; mkdir -p "$OUT"
SETS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
k=0
for S in "${SETS[@]}"; do
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $S --output-format csv -d "$OUT/set$k" -- python3 $CMD > "$OUT/set$k.log" 2>&1 < /dev/null || echo "set $k failed: $(tail -2 "$OUT/set$k.log" | cut -c1-200)"
  k=$((k+1))
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$OUT/set*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "step_kernel" not in kn: continue
        key = kn.split("(")[0][-48:]
        a = agg[key][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for key, d in agg.items():
    print("==", key)
    for c, (n, tot) in sorted(d.items()):
        print("  %-34s launches %4d  mean/launch %16.1f   per tile %10.2f" % (c, n, tot / n, tot / n / 156250.0))
PY
find "$OUT" -name "*.csv" -size +1M -delete
