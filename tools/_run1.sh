cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python tools/sort_decay.py 12 2>&1 | grep field
