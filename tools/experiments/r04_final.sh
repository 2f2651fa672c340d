# round 4, final build: the numbers quoted in DESIGN.md section 7 (one box)
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04f_bench.jsonl
python bench.py 2>/dev/null | tail -1 >> gpurun_out/r04f_bench.jsonl
python bench.py --force-dist 2>/dev/null | tail -1 >> gpurun_out/r04f_bench.jsonl
: > gpurun_out/r04f_cases.jsonl
for o in box_records=1 box_records=0; do
  for c in "box3d --field diagonal" "box3d --field swirl" "tjunction --field u0=3" "tjunction --field u0=5" "box3d --field swirl --D 1.5e-5" "tjunction --field u0=3 --D 1.5e-5" "tjunction_run --particles 4e6 --D 1.5e-5"; do
    timeout -s KILL 200 python tools/bench_case.py --case $c --opt $o --label $o 2>/dev/null | tail -1 >> gpurun_out/r04f_cases.jsonl
  done
  for n in 1.25e6 1e7 1e8; do
    CPF_BOX_N=128,128,128 timeout -s KILL 300 python tools/bench_case.py --case box3d --field diagonal --particles $n --opt $o --label "$o 2.1e6 cells" 2>/dev/null | tail -1 >> gpurun_out/r04f_cases.jsonl
  done
done
