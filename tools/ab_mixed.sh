#!/usr/bin/env bash
# GPU box: tools/bench_mixed.py once per prebuilt library build_ab/lib_<name>.so (same box, alternating)
cd "$(dirname "$0")/.." || exit 1
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so
for rep in 1 2; do
  for f in build_ab/lib_*.so; do
    name=$(basename "$f" .so); name=${name#lib_}
    cp "$f" $LIB
    python tools/bench_mixed.py 2>/dev/null | python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("'$name'", d["kernel_ms"], d["kernel"].split("<")[1], d["case"][:60])'
  done
done
cp /tmp/lib_orig.so $LIB
