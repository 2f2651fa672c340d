cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T="timeout -s KILL 300"
$T python tools/sweep.py --variants 3,4 --no-stats --steps 30 2>&1 | grep '"variant'
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so; cp $LIB /tmp/lib_orig.so
for L in g256 g64 g32 g8; do cp build_ab/lib_$L.so $LIB; 
  for O in "stream_tail_fraction=0.1" "stream_tail_fraction=0.05"; do
  $T python tools/stream_timeline.py --label $L --opt $O 2>&1 | grep kernel_ms | cut -c1-330
  done
done
cp /tmp/lib_orig.so $LIB
