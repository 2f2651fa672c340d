# round 6, final build (the streaming kernel's body refactored into stream_body<..., VERTEX>): the randomised parity campaign
# (tools/fuzz_parity.py), all modes, fresh seeds; then the shard logic through the HIP kernels and the vertex tests
for m in "" mixed poly box flat; do
  echo "== tools/fuzz_parity.py 150000 N $m (round 6, final build)"
  case "$m" in box) N=8000;; flat) N=6000;; "") N=6000;; *) N=3000;; esac
  timeout -s KILL 1500 python tools/fuzz_parity.py 150000 $N $m 2>&1 | tail -2
done
echo "== CPF_FUZZ_BLOCKS=40 tests/test_gpu_shard_fuzz.py"
CPF_FUZZ_BLOCKS=40 timeout -s KILL 1500 python -m pytest tests/test_gpu_shard_fuzz.py -x -q 2>&1 | tail -2
