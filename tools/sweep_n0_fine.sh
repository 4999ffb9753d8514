#!/bin/bash
# usage (GPU box): bash tools/sweep_n0_fine.sh -- 16M sphere, rows per partition around the default (975 at k = 7, 900 at k = 8), 700 steps each
set -o pipefail
b() { timeout -k 10 250 python bench.py --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  %.1f us' % (d['value']/1e9, 1e3*d['roofline']['avg_launch_ms']), d['config']['tiles']['ntiles'], d['config']['tiles']['max_local'])"; }
for kn in ${SWEEP_KN:-7:975 7:940 7:1000 7:1024 7:1050 7:1100 8:900 8:940 7:975}; do
  k=${kn%%:*}; n0=${kn##*:}
  echo "k=$k n0=$n0  $(FEM_TILE_N0=$n0 b --fuse $k)" || exit 1
done
