#!/usr/bin/env bash
# GPU box: run-time options of the streaming kernel / the sort key on the 3-D bench meshes (tools/bench_3d.py);
# one option set per line of $SWEEP_FILE (default: the dealing knobs)
cd "$(dirname "$0")/.." || exit 1
DEFAULT=$'\nstream_tiles_per_chunk=2 stream_tail_fraction=0.2\nstream_tiles_per_chunk=2 stream_tail_fraction=0.3\nstream_tiles_per_chunk=1\nstream_tiles_per_chunk=3 stream_tail_fraction=0.2\nstream_tiles_per_chunk=2 stream_tail_fraction=0.1'
while IFS= read -r opts; do
  echo "== [$opts] $(CPF_OPTS="$opts" python tools/bench_3d.py "$@" 2>&1 | grep kernel_ms | python -c 'import sys,json
print(" ".join(str(json.loads(l)["kernel_ms"]) for l in sys.stdin))')"
done <<< "${SWEEP_SETS:-$DEFAULT}"
