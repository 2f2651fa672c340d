AB_TIMEOUT=150 tools/ab.sh -r 1 -- python tools/bench_case.py --case box3d --field diagonal
AB_TIMEOUT=150 tools/ab.sh -r 1 -- python tools/bench_case.py --case box3d --field swirl
AB_TIMEOUT=150 tools/ab.sh -r 1 -- python tools/bench_case.py --case tjunction --field u0=3
AB_TIMEOUT=150 tools/ab.sh -r 1 -- python tools/bench_case.py --case tjunction_run --particles 4e6 --D 1.5e-5
AB_TIMEOUT=150 tools/ab.sh -r 1 -- python tools/bench_case.py --case box3d --field swirl --D 1.5e-5
