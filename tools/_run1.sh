cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 400 python bench.py > gpurun_out/r02_bench_1gpu.json 2> gpurun_out/r02_bench_1gpu.err; cut -c1-200 gpurun_out/r02_bench_1gpu.json
timeout -s KILL 700 bash tools/profile_run.sh r02 > gpurun_out/r02_profile_run.log 2>&1; tail -2 gpurun_out/r02_profile_run.log | cut -c1-100
timeout -s KILL 600 bash tools/profile_3d.sh r02 2>&1 | grep kernel_ms | cut -c1-200
CPF_TJUNCTION=1 timeout -s KILL 300 python tools/bench_3d.py 2>&1 | grep kernel_ms | cut -c1-200
timeout -s KILL 300 python tools/bench_pimple.py 2>&1 | tail -1 | cut -c1-300
