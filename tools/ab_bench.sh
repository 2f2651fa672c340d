#!/usr/bin/env bash
# GPU box: like tools/ab.sh, but runs bench.py (whole step loop incl. the periodic sort) for every build_ab/lib_*.so
cd "$(dirname "$0")/.." || exit 1
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so
for rep in 1 2; do
  for f in build_ab/lib_*.so; do
    name=$(basename "$f" .so); name=${name#lib_}
    cp "$f" $LIB
    echo "$name: $(python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep metric | python -c 'import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print(d["value"], d["ms_per_step"], "kernel", d["roofline"]["kernel_avg_ms"], "frac", d["roofline"]["frac"], "steady", (c["ms_per_step_steady"] or {}).get("ms_per_step"), "brownian", (c["brownian"] or {}).get("kernel_avg_ms"), "fused8", (c["extra_fused_cycles"] or {}).get("ms_per_cycle"))')"
  done
done
cp /tmp/lib_orig.so $LIB
