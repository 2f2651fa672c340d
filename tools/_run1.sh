cd $GRAFT_REPO_ROOT
tools/ab2.sh --variants 4 --no-floor 2>&1 | grep variant
