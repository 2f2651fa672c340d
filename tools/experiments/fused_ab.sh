for st in 8 16; do
  for o in step_variant=-1 step_variant=4 step_variant=3; do
    timeout -s KILL 300 python tools/bench_case.py --case pitz --field uniform --fused --steps $st --opt $o --label "fused$st $o" 2>/dev/null | tail -1
  done
done
for o in step_variant=-1 step_variant=4; do
  timeout -s KILL 300 python tools/bench_case.py --case pitz --field uniform --fused --steps 8 --D 1.5e-5 --opt $o --label "fused8 D $o" 2>/dev/null | tail -1
  timeout -s KILL 300 python tools/bench_case.py --case tjunction --field u0=3 --fused --steps 8 --opt $o --label "tj fused8 $o" 2>/dev/null | tail -1
done
