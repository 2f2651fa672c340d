# where the loop lookup (0) overtakes the fixed tag compare (1) on pitzDaily (12 225 cells): particles per cell 41 ... 818
for n in 5e5 1e6 2e6 4e6 1e7; do
  for o in stream_lookup=0 stream_lookup=1; do
    timeout -s KILL 300 python tools/bench_case.py --case pitz --field uniform --particles $n --opt $o --label "n=$n $o" 2>/dev/null | tail -1
  done
done
