#!/usr/bin/env bash
# per-kernel averages of tools/sort_timing.py under rocprofv3 for each A/B library named on the command line
cd /root/repo
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so
export TMPDIR=/tmp
for name in "$@"; do
  if [ "$name" = product ]; then cp /tmp/lib_orig.so $LIB; else cp build_ab/lib_$name.so $LIB; fi
  rm -rf /tmp/prof_$name
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 /root/repo/tools/sort_timing.py > /dev/null 2>&1)
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name"; python3 tools/scratch/stats.py $f "rt_scatter|rt_hist|rt_scan"
done
cp /tmp/lib_orig.so $LIB
