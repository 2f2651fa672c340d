cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests/test_compat_gpu.py -x -q -k tjunction 2>&1 | tail -15
