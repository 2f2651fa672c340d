cd $GRAFT_REPO_ROOT
timeout -s KILL 1500 python tools/fuzz_parity.py 10000 30000 2>&1 | tail -4
