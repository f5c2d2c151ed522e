BENCH="python bench.py --steps 20 --warmup 5"
for o in "" "tile_cells=64" "tile_conv=0" "fuse_gru=0" "tile_volume=0" "fuse_lookup=0" "fuse_flow=0" "fuse_ou=0" ""; do
   args=""; for kv in $o; do args="$args --engine-opt $kv"; done
   $BENCH --no-cpu-baseline --no-parity --no-alt-arith --no-host-io --no-profile $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('engine options [$o]:', round(d['value'],1), 'frames/s')"
done
