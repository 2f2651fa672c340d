#!/usr/bin/env bash
# GPU box: A/B of two prebuilt libraries (build_ab/lib_prev.so, build_ab/lib_new.so) inside ONE gpurun call
# (every call lands on a different MI355X, so only same-call comparisons mean anything).  Alternates A B A B.
cd "$(dirname "$0")/.." || exit 1
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so
for rep in 1 2; do
  for which in prev new; do
    cp build_ab/lib_$which.so $LIB
    echo "$which: $(python tools/sweep.py --variants 3 --no-stats --steps 20 "$@" 2>/dev/null | grep 'variant": 3')"
  done
done
cp /tmp/lib_orig.so $LIB
