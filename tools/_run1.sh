cd $GRAFT_REPO_ROOT
timeout -s KILL 400 python bench.py --no-cpu-baseline --force-dist --particles 2e5 --steps 6 --warmup 2 > /tmp/out.txt 2> /tmp/err.txt
echo "--- stdout lines: $(wc -l < /tmp/out.txt)"; cut -c1-100 /tmp/out.txt
timeout -s KILL 400 python bench.py --no-cpu-baseline --force-dist 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], c['ms_in_handoff'], c['handoff_fraction_per_step'], c['exchange_interval'], c['rebalance_interval'], c['overlap_steps'])"
