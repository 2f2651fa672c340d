cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
T="timeout -s KILL 300"
for V in 3 4; do CPF_VARIANT=$V $T python tools/bench_3d.py 2>&1 | grep kernel_ms; done
CPF_VARIANT=4 CPF_OPTS="stream_tiles_per_chunk=2" $T python tools/bench_3d.py 2>&1 | grep kernel_ms
for V in 3 4; do CPF_VARIANT=$V $T python tools/bench_pimple.py 2>&1 | tail -3; done
