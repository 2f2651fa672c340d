cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PMC_CMD="tools/sweep.py --no-floor --spinup-ms 0 --variants 4 --steps 20 --warmup 2 --no-stats" timeout -s KILL 700 bash tools/pmc_3d.sh pitz > gpurun_out/r02_sq_pitz.txt 2>&1
timeout -s KILL 700 bash tools/pmc_3d.sh m3d > gpurun_out/r02_sq_3d.txt 2>&1
CPF_TJUNCTION=1 timeout -s KILL 700 bash tools/pmc_3d.sh tj > gpurun_out/r02_sq_tj.txt 2>&1
grep -c "mean/launch" gpurun_out/r02_sq_pitz.txt gpurun_out/r02_sq_3d.txt gpurun_out/r02_sq_tj.txt
