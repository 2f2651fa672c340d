#!/usr/bin/env bash
# GPU box: A/B/... of prebuilt libraries build_ab/lib_<name>.so inside ONE gpurun call (every call lands on a
# different MI355X, so only same-call comparisons mean anything).  Two alternating passes over all of them.
cd "$(dirname "$0")/.." || exit 1
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so
for rep in 1 2; do
  for f in build_ab/lib_*.so; do
    name=$(basename "$f" .so); name=${name#lib_}
    cp "$f" $LIB
    echo "$name: $(python tools/sweep.py --variants 3 --no-stats --steps 20 "$@" 2>/dev/null | grep 'variant": 3')"
  done
done
cp /tmp/lib_orig.so $LIB
