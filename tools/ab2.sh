#!/usr/bin/env bash
# GPU box: runs tools/sweep.py with the given arguments once per prebuilt library build_ab/lib_<name>.so, two
# alternating passes (every gpurun call lands on a different MI355X: only same-call comparisons mean anything).
cd "$(dirname "$0")/.." || exit 1
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so
for rep in 1 2; do
  for f in build_ab/lib_*.so; do
    name=$(basename "$f" .so); name=${name#lib_}
    cp "$f" $LIB
    python tools/sweep.py --no-stats --steps 20 --label "$name" "$@" 2>&1 | grep '"variant"' | grep -v zero-cycle
  done
done
cp /tmp/lib_orig.so $LIB
