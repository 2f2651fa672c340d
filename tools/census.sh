#!/usr/bin/env bash
# GPU box: how many waves of the streaming kernel are really resident?  Timeline build (build_ab/lib_onetl.so,
# -DCPF_STREAM_TIMELINE), grid sized for 24 / 28 / 32 single-wave workgroups per CU: a wave slot the hardware does not
# admit starts only when another wave has exited (start time >> 0).
cd "$(dirname "$0")/.." || exit 1
LIB=cudaparticlesfoam_amd/lib/libcudaParticleAdvection.so
cp $LIB /tmp/lib_orig.so
cp build_ab/lib_onetl.so $LIB
for w in 24 28 32; do
  python tools/stream_timeline.py --label "waves_per_cu=$w" --opt stream_waves_per_cu=$w "$@" 2>&1 | grep kernel_ms
done
cp /tmp/lib_orig.so $LIB
