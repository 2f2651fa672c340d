# box meshes with MANY particles per cell: loop lookup on 256-byte records (what >= 128 per cell picked) against box records
for nb in 16,16,8 32,32,20 48,40,32; do
  for o in stream_lookup=0 stream_lookup=6; do
    CPF_BOX_N=$nb timeout -s KILL 300 python tools/bench_case.py --case box3d --field swirl --opt $o --label "$nb $o" 2>/dev/null | tail -1
    CPF_BOX_N=$nb timeout -s KILL 300 python tools/bench_case.py --case box3d --field swirl --D 1.5e-5 --opt $o --label "$nb $o D" 2>/dev/null | tail -1
  done
done
