#!/usr/bin/env bash
# GPU box: A/B of prebuilt libraries inside ONE gpurun call (every call lands on a different MI355X and the boxes differ by
# up to 8 %: only same-call comparisons mean anything).
#   tools/ab.sh [-r REPS] [-l "name1 name2 ..."] -- <command ...>
# runs the command REPS times (default 2, alternating) per build_ab/lib_<name>.so (all of them unless -l names some;
# "product" = the library in place) and prefixes every JSON line of its output with the library's name.
# Libraries come from tools/build_ab.sh.  Every run of the command is under `timeout` (AB_TIMEOUT seconds, default 300): an
# experimental kernel that never finishes must cost its own limit, not the call's.
cd "$(dirname "$0")/.." || exit 1
REPS=2; NAMES=""
while [ $# -gt 0 ]; do
  case "$1" in -r) REPS=$2; shift 2;; -l) NAMES=$2; shift 2;; --) shift; break;; *) break;; esac
done
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so
[ -z "$NAMES" ] && NAMES=$(for f in build_ab/lib_*.so; do n=$(basename "$f" .so); echo "${n#lib_}"; done)
for rep in $(seq 1 "$REPS"); do
  for name in $NAMES; do
    if [ "$name" = product ]; then cp /tmp/lib_orig.so $LIB; else cp "build_ab/lib_$name.so" $LIB || continue; fi
    timeout -s KILL ${AB_TIMEOUT:-300} "$@" 2>/tmp/ab_err.log | grep '^{' | sed "s/^/$name: /" || tail -3 /tmp/ab_err.log
  done
done
cp /tmp/lib_orig.so $LIB
