#!/bin/bash
# usage (GPU box): bash tools/sweep_scan_cfg8.sh -- scan8m on the 2048-row register tile (FEM_TILE_CONFIG=8) over fused depth x rows per partition
set -o pipefail
b() { timeout -k 10 250 python bench.py --workload scan8m --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  %.1f us' % (d['value']/1e9, 1e3*d['roofline']['avg_launch_ms']), d['config']['tiles']['ntiles'], d['config']['tiles']['max_local'], d['config']['tiles']['threads'], d['config']['tiles']['rows_per_thread'])"; }
echo "default     $(b)"
for k in 6 7; do
  for n0 in ${SWEEP_N0:-800 900 1000}; do
    echo "cfg8 k=$k n0=$n0  $(FEM_TILE_CONFIG=8 FEM_TILE_N0=$n0 b --fuse $k)" || exit 1
  done
done
echo "cfg8 k=5 n0=1100  $(FEM_TILE_CONFIG=8 FEM_TILE_N0=1100 b --fuse 5)"
