#!/usr/bin/env bash
# GPU box: the streaming kernel's dealing knobs (tiles per chunk, share of the cloud dealt tile by tile at the end) on one box
cd "$(dirname "$0")/.." || exit 1
for rep in ${REPS:-1 2}; do
for tf in ${TFS:-0.1 0.2 0.3 0.5}; do
  for tpc in ${TPCS:-2 4 8}; do
    echo "tail=$tf tpc=$tpc: $(python tools/sweep.py --no-stats --steps 20 --variants 4 --no-floor --opt stream_tail_fraction=$tf --opt stream_tiles_per_chunk=$tpc 2>&1 | grep '"variant"' | python -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d.get("kernel_ms", d))')"
  done
done
done
