# round 4, final build: the randomised parity campaign (tools/fuzz_parity.py), all modes
for m in "" mixed poly box flat; do
  echo "== tools/fuzz_parity.py 40000 N $m (final build)"
  case "$m" in box) N=4000;; flat) N=1500;; "") N=3000;; *) N=1500;; esac
  timeout -s KILL 900 python tools/fuzz_parity.py 40000 $N $m 2>&1 | tail -2
done
