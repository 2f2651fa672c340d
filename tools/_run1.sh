cd $GRAFT_REPO_ROOT
CPF_FUZZ_VARIANT=5 timeout -s KILL 600 python tools/fuzz_parity.py 200000 4000 2>&1 | tail -1
CPF_FUZZ_VARIANT=3 timeout -s KILL 600 python tools/fuzz_parity.py 300000 4000 2>&1 | tail -1
CPF_FUZZ_VARIANT=0 timeout -s KILL 600 python tools/fuzz_parity.py 400000 2000 2>&1 | tail -1
timeout -s KILL 900 python tools/fuzz_parity.py 500000 20000 2>&1 | tail -1
