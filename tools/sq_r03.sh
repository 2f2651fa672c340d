#!/usr/bin/env bash
# GPU box: the round's SQ counter sections (per 64-particle tile) for the headline, the tutorial diffusion and the 3-D mesh
# -> gpurun_out/r03_sq_counters.txt; rounds per tile from the timeline build (build_ab/lib_onetl.so).
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r03_sq_counters.txt
{
echo "# SQ counters of the statistics-off step kernel, per launch and per 64-particle tile (1e7 particles = 156 250 tiles)."
echo "# tools/pmc_3d.sh with PMC_CMD (pitzDaily sweeps) / default (3-D bench mesh); round 3 kernels."
echo "## pitz (pitzDaily, D = 0, variant 4)"
PMC_CMD="tools/sweep.py --variants 4 --no-stats --no-floor --spinup-ms 0 --steps 6 --warmup 2" timeout 900 bash tools/pmc_3d.sh pitz 2>&1 | grep -v Warn
echo "## brown (pitzDaily, D = 1.5e-5, variant 4)"
PMC_CMD="tools/sweep.py --variants 4 --no-stats --no-floor --spinup-ms 0 --steps 6 --warmup 2 --D 1.5e-5" timeout 900 bash tools/pmc_3d.sh brown 2>&1 | grep -v Warn
echo "## 3d (245 760-cell graded box, diagonal + swirl fields)"
timeout 900 bash tools/pmc_3d.sh 3d 2>&1 | grep -v Warn
} > $OUT
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so; cp build_ab/lib_onetl.so $LIB
{
echo "## rounds per tile and wave timeline (timeline build, cold clocks)"
python tools/stream_timeline.py --label pitz 2>&1 | grep kernel_ms
python tools/stream_timeline.py --label brown --D 1.5e-5 2>&1 | grep kernel_ms
python tools/stream_timeline.py --label 3d --mesh3d 2>&1 | grep kernel_ms
} >> $OUT
cp /tmp/lib_orig.so $LIB
tail -5 $OUT | cut -c1-400
