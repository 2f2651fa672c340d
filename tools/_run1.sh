cd $GRAFT_REPO_ROOT
T="timeout -s KILL 600"
$T python -m pytest tests/test_gpu_parity.py -x -q -k "lookup or ragged or bit_exact" 2>&1 | tail -4
CPF_CHECK_LOOKUP=1 $T python tools/stream_check.py 2>&1 | tail -1
CPF_CHECK_LOOKUP=0 $T python tools/stream_check.py 2>&1 | tail -1
for i in 1 2; do
$T python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], 'brownian', d['config']['brownian']['kernel_avg_ms'])"
CPF_VARIANT=4 $T python tools/bench_3d.py 2>&1 | grep kernel_ms | cut -c1-120
CPF_TJUNCTION=1 $T python tools/bench_3d.py 2>&1 | grep kernel_ms | cut -c1-130
$T python tools/bench_pimple.py 2>&1 | tail -1 | cut -c100-200
done
