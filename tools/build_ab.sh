#!/usr/bin/env bash
# Builds A/B variants of the library that differ only in the -D flags of ONE translation unit:
#   tools/build_ab.sh cpf_stream.hip name1 "-DFLAG=1" name2 "-DFLAG=2 -DOTHER" ...
# -> build_ab/lib_<name>.so (git-ignored; travels to the GPU box).  Run them with tools/ab.sh.
set -e
cd "$(dirname "$0")/.."
SRC=$1; shift
CS=cudaparticlesfoam_amd/csrc
make -C $CS -s -j4
mkdir -p build_ab
BASE=$(basename $SRC .hip)
OTHERS=$(ls $CS/build/*.o | grep -v "/$BASE.o")
while [ $# -gt 1 ]; do
  NAME=$1; FLAGS=$2; shift 2
  SF=""; [ "$BASE" = "cpf_stream" ] && SF="-mllvm --amdgpu-sched-strategy=max-ilp"      # as in csrc/Makefile
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off $SF -Iinclude -I$CS $FLAGS -c $CS/$SRC -o /tmp/ab_$NAME.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/ab_$NAME.o -o build_ab/lib_$NAME.so
  echo "built build_ab/lib_$NAME.so ($FLAGS)"
done
