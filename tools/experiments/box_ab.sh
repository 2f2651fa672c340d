AB_TIMEOUT=150 tools/ab.sh -r 2 -- python tools/bench_case.py --case tjunction_run --particles 4e6 --D 1.5e-5
AB_TIMEOUT=150 tools/ab.sh -r 1 -- python tools/bench_case.py --case tjunction --field u0=3 --D 1.5e-5
AB_TIMEOUT=150 tools/ab.sh -r 1 -- python tools/bench_case.py --case box3d --field swirl --D 1e-5
