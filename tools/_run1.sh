cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests/test_gpu_tjunction.py -x -q 2>&1 | tail -5
CPF_TJUNCTION=1 timeout -s KILL 300 python tools/bench_3d.py 2>&1 | grep kernel_ms | cut -c1-250
