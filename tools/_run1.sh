cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python -m pytest tests/test_gpu_goldens.py -m gpu -x -q 2>&1 | tail -8
