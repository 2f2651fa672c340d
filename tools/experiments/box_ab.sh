set -x
timeout -s KILL 900 python -m pytest tests/test_gpu_box.py -x -q 2>&1 | tail -15
AB_TIMEOUT=120 tools/ab.sh -r 1 -- python tools/bench_case.py --case box3d --field diagonal
AB_TIMEOUT=120 tools/ab.sh -r 1 -- python tools/bench_case.py --case box3d --field swirl
AB_TIMEOUT=120 tools/ab.sh -r 1 -- python tools/bench_case.py --case tjunction --field u0=3
export CPF_BOX_N=128,128,128
for o in "box_records=0" "box_records=1"; do
  for n in 1.25e6 1e7; do
    timeout -s KILL 200 python tools/bench_case.py --case box3d --field diagonal --particles $n --opt $o --label "$o n=$n" 2>&1 | tail -1
  done
done
