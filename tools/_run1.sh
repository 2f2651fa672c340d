cd $GRAFT_REPO_ROOT
T="timeout -s KILL 300"
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so; cp $LIB /tmp/lib_orig.so
cp build_ab/lib_efs.so $LIB; CPF_CHECK_VARIANT=4 $T python tools/stream_check.py 2>&1 | tail -1
for L in base efs base efs; do cp build_ab/lib_$L.so $LIB
echo "== $L"
$T python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], 'brownian', d['config']['brownian']['kernel_avg_ms'])"
CPF_VARIANT=4 $T python tools/bench_3d.py 2>&1 | grep kernel_ms | cut -c1-120
CPF_TJUNCTION=1 $T python tools/bench_3d.py 2>&1 | grep kernel_ms | cut -c1-130
done
cp /tmp/lib_orig.so $LIB
