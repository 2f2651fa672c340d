cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gputest.log 2>&1; grep -E "passed|failed|error" gpurun_out/r02_gputest.log | tail -3
