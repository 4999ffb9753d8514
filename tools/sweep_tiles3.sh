#!/bin/bash
# usage (GPU box): bash tools/sweep_tiles3.sh OUT "cfg:n0:k ..." [workload]   -- one bench line per tile shape
OUT=$1; WL=${3:-sphere16m}
: > $OUT
for spec in $2; do
  IFS=: read cfg n0 k <<< "$spec"
  v=$(FEM_TILE_CONFIG=$cfg FEM_TILE_N0=$n0 timeout -k 10 300 python bench.py --workload $WL --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 --warmup 8 --steps 600 --fuse $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G  rows_loaded %d ntiles %d' % (d['value']/1e9, d['config']['tiles']['rows_loaded'], d['config']['tiles']['ntiles']))" 2>/dev/null) || v=failed
  echo "cfg=$cfg n0=$n0 k=$k $v" >> $OUT
done
cat $OUT
