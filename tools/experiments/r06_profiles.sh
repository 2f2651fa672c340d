#!/usr/bin/env bash
# GPU box, ONE gpurun call: the round-6 profiles (tools/pmc.sh per label: rocprofv3 --kernel-trace --stats, then separate --pmc
# FETCH_SIZE / WRITE_SIZE passes; --sq: the SQ counter passes).  Every label runs bench.py with everything off except what it
# profiles, so that a kernel's per-launch average in the trace is that workload's.  Afterwards, here:
#   python tools/condense_profile.py r06 --bench; python tools/condense_profile.py r06_brown --algo-bytes 640000000 --match "<true"; ...
cd "$(dirname "$0")/../.." || exit 1
OFF="--no-cpu-baseline --anchor-particles 0 --brownian-extra 0 --fused-extra 0 --steady-steps 0 --brownian-steady-steps 0 --analytic-extra 0 --tjunction-steps 0 --vertex-steps 0 --polyhedral-steps 0"
off() { echo "$OFF" | sed "s/$1 0//"; }
# the headline alone: the 100 launches of the statistics-off instantiation are exactly the timed steps (raw trace kept)
tools/pmc.sh r06 -- python3 bench.py --steps 100 --warmup 10 $OFF
cp $(find gpurun_out/r06/stats -name "*kernel_trace.csv" | head -1) gpurun_out/r06_headline_kernel_trace.csv 2>/dev/null
tools/pmc.sh r06_brown --sq -- python3 bench.py --steps 20 --warmup 5 $(off --brownian-steady-steps) --brownian-steady-steps 100 --no-fused-tutorial
tools/pmc.sh r06_vertex -- python3 bench.py --steps 20 --warmup 5 $(off --vertex-steps) --vertex-steps 50
tools/pmc.sh r06_poly -- python3 bench.py --steps 20 --warmup 5 $(off --polyhedral-steps) --polyhedral-steps 50
tools/pmc.sh r06_tjrun -- python3 bench.py --steps 20 --warmup 5 $(off --tjunction-steps) --tjunction-steps 100 --no-fused-tutorial
