timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -k "flat or lookup_method" 2>&1 | tail -4
for o in flat_walk=1 flat_walk=0; do
  for f in uniform analytic; do
  timeout -s KILL 300 python tools/bench_case.py --case pitz --field $f --opt $o --label "$o" 2>/dev/null | tail -1
  done
done
for o in flat_walk=1 flat_walk=0; do
  for f in uniform analytic; do
  timeout -s KILL 300 python tools/bench_case.py --case pitz --field $f --opt $o --label "$o" 2>/dev/null | tail -1
  done
done
