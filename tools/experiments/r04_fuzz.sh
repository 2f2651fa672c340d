# round 4, final build: the randomised parity campaign (tools/fuzz_parity.py), all modes
for m in "" mixed poly box flat; do
  echo "== tools/fuzz_parity.py 60000 N $m (final build)"
  case "$m" in box) N=12000;; flat) N=8000;; "") N=8000;; *) N=4000;; esac
  timeout -s KILL 1500 python tools/fuzz_parity.py 60000 $N $m 2>&1 | tail -2
done
