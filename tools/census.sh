#!/usr/bin/env bash
# GPU box: census of the streaming kernel's rounds (timeline build build_ab/lib_onetl.so, -DCPF_STREAM_TIMELINE): busy lanes
# per round index, sit-outs, rounds per tile against the largest visit count -> gpurun_out/<label>_census.jsonl
cd "$(dirname "$0")/.." || exit 1
LABEL="${1:-r04}"
OUT=gpurun_out/${LABEL}_census.jsonl; : > $OUT
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so
cp build_ab/lib_onetl.so $LIB
run() { timeout -s KILL 300 python tools/stream_timeline.py --census "$@" > /tmp/census_run.log 2>&1; grep kernel_ms /tmp/census_run.log >> $OUT || tail -5 /tmp/census_run.log; }
if [ -n "$CENSUS_CASES" ]; then
  for cs in $CENSUS_CASES; do run --label "$cs" --case "$cs"; done
else
  run --label pitz
  run --label pitz_brown --D 1.5e-5
  run --label box3d --mesh3d
  run --label box3d_brown --mesh3d --D 1.5e-5
  run --label tjunction --tjunction
  run --label tjunction_brown_4e6 --tjunction --D 1.5e-5 --particles 4e6
fi
cp /tmp/lib_orig.so $LIB
cut -c1-300 $OUT
