cd $GRAFT_REPO_ROOT
PMC_CMD="tools/sweep.py --no-floor --spinup-ms 0 --variants 4 --steps 20 --warmup 2 --no-stats --D 1.5e-5" timeout -s KILL 700 bash tools/pmc_3d.sh brown > gpurun_out/r02_sq_brown.txt 2>&1
grep -E "^==|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_INSTS_BRANCH|SQ_INSTS_LDS|SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_ACTIVE_INST_VALU|SQ_INSTS_VMEM" gpurun_out/r02_sq_brown.txt | cut -c1-110
