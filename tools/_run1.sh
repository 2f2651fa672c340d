cd $GRAFT_REPO_ROOT
timeout -s KILL 900 tools/profile_run.sh r02 2>&1 | tail -30
timeout -s KILL 600 tools/profile_3d.sh r02 2>&1 | tail -30
