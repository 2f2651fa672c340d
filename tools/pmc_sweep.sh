#!/usr/bin/env bash
# GPU box: SQ counter passes over tools/sweep.py for the given variants; prints per-kernel means.
# usage: tools/pmc_sweep.sh "2,4" [extra sweep args]
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
VARS="${1:-3}"; shift || true
OUT=gpurun_out/pmc_sweep; rm -rf "$OUT"; mkdir -p "$OUT"
SETS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
k=0
for S in "${SETS[@]}"; do
  rocprofv3 --kernel-trace --pmc $S --output-format csv -d "$OUT/set$k" -- python3 tools/sweep.py --no-floor --spinup-ms 0 --variants "$VARS" --steps 6 --warmup 2 "$@" > "$OUT/set$k.log" 2>&1
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
    waves = d.get("SQ_WAVES", [1, 1.0]); w = waves[1] / max(waves[0], 1)
    for c, (n, tot) in sorted(d.items()):
        print("  %-28s mean/launch %14.1f   per-wave %10.2f" % (c, tot / n, tot / n / max(w, 1)))
PY
