#!/usr/bin/env bash
# headline value for event strides 4 and 20, alternating, REPS times (run under tools/ab.sh for library variants)
OFF="--no-cpu-baseline --brownian-extra 0 --fused-extra 0 --steady-steps 0 --anchor-particles 0 --analytic-extra 0 --brownian-steady-steps 0 --tjunction-steps 0"
for r in 1 2 3; do for s in 4 20; do
  python bench.py $OFF --timing-stride $s 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'stride': $s, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'kernel_avg_ms': d['roofline']['kernel_avg_ms']}))"
done; done
