cd $GRAFT_REPO_ROOT
T="timeout -s KILL 300"
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so; cp $LIB /tmp/lib_orig.so
CPF_CHECK_VARIANT=4 $T python tools/stream_check.py 2>&1 | tail -2
for L in t0 t2 t3 t5 t0 t2 t3; do cp build_ab/lib_$L.so $LIB
$T python tools/stream_timeline.py --label $L --groups 2>&1 | grep kernel_ms | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['label'], d['kernel_ms'], 'end', d['end_us_pct'], 'grp_last', d['group_last_end_pct'], 'busy', d['slot_busy_fraction'])"
done
cp build_ab/lib_t2.so $LIB
for o in 0.05 0.2 0.3; do $T python tools/stream_timeline.py --label t2_tail$o --opt stream_tail_fraction=$o --groups 2>&1 | grep kernel_ms | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['label'], d['kernel_ms'], 'end', d['end_us_pct'], 'grp_last', d['group_last_end_pct'], 'busy', d['slot_busy_fraction'])"
done
cp /tmp/lib_orig.so $LIB
