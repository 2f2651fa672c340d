cd $GRAFT_REPO_ROOT
T="timeout -s KILL 300"
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so; cp $LIB /tmp/lib_orig.so
cp build_ab/lib_tl.so $LIB
for B in 250 700 520 430 610 340 160; do for P in 5 60; do
  $T python tools/stream_timeline.py --label "bits$B pre$P" --pre-steps $P --opt sort_key_bits=$B 2>&1 | grep kernel_ms | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['label'], d['kernel_ms'], 'rounds/tile', d['rounds_per_tile'], 'end', d['end_us_pct'][3], d['end_us_pct'][6])"
done; done
cp /tmp/lib_orig.so $LIB
