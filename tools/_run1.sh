cd $GRAFT_REPO_ROOT
T="timeout -s KILL 900"
CPF_CHECK_LOOKUP=1 $T python tools/stream_check.py 2>&1 | tail -1
CPF_CHECK_LOOKUP=0 $T python tools/stream_check.py 2>&1 | tail -1
timeout -s KILL 600 python tools/fuzz_parity.py 70000 3000 2>&1 | tail -1
timeout -s KILL 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gputest.log 2>&1; grep -E "passed|failed|error" gpurun_out/r02_gputest.log | tail -3
