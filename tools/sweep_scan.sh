#!/bin/bash
# usage (GPU box): bash tools/sweep_scan.sh -- scan8m over fused depth x rows per partition (700 steps each)
set -o pipefail
b() { timeout -k 10 250 python bench.py --workload scan8m --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  %.1f us' % (d['value']/1e9, 1e3*d['roofline']['avg_launch_ms']), d['config']['tiles']['ntiles'], d['config']['tiles']['max_local'])"; }
for k in 5 6; do
  for n0 in ${SWEEP_N0:-760 820 880 940}; do
    echo "k=$k n0=$n0  $(FEM_TILE_N0=$n0 b --fuse $k)" || exit 1
  done
done
