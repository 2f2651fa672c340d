#!/usr/bin/env bash
# GPU box: tools/bench_3d.py once per prebuilt library build_ab/lib_<name>.so (optionally only those named in $AB_LIBS)
cd "$(dirname "$0")/.." || exit 1
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so
for f in build_ab/lib_*.so; do
  name=$(basename "$f" .so); name=${name#lib_}
  if [ -n "$AB_LIBS" ] && ! echo " $AB_LIBS " | grep -q " $name "; then continue; fi
  cp "$f" $LIB
  python tools/bench_3d.py "$@" 2>&1 | grep kernel_ms | sed "s/^/$name: /"
done
cp /tmp/lib_orig.so $LIB
