for n in 3e5 1.2e6 2.4e6; do
  for o in stream_lookup=3 stream_lookup=7; do
  CPF_BOX_N=64,64,64 timeout -s KILL 400 python tools/bench_case.py --case refbox3d --field diagonal --particles $n --opt $o --label "refined $o n=$n" 2>/dev/null | tail -1
  done
done
timeout -s KILL 600 python -m pytest tests/test_gpu_mixed.py -x -q 2>&1 | tail -2
