timeout -s KILL 600 python tools/fuzz_parity.py 30000 300 flat > gpurun_out/fuzz_flat.txt 2>&1; tail -1 gpurun_out/fuzz_flat.txt
tools/pmc.sh r04 -- python3 bench.py --no-cpu-baseline > gpurun_out/pmc_bench.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04g_bench.jsonl
python bench.py 2>/dev/null | tail -1 >> gpurun_out/r04g_bench.jsonl
python bench.py --force-dist 2>/dev/null | tail -1 >> gpurun_out/r04g_bench.jsonl
python bench.py --field analytic --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r04g_bench.jsonl
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
