timeout -s KILL 1500 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.txt 2>&1; grep -E "passed|failed|error|FAILED" gpurun_out/gpu_tests.txt | tail -6
timeout -s KILL 300 python tools/fuzz_parity.py 31000 200 flat 2>&1 | tail -1
for o in flat_walk=1 flat_walk=0; do
  timeout -s KILL 300 python tools/bench_case.py --case pitz --field uniform --particles 1e6 --opt $o --label "$o 1e6" 2>/dev/null | tail -1
  CPF_OPTS="$o" timeout -s KILL 300 python tools/bench_pimple.py 2>/dev/null | tail -1
done
