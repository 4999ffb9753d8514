#!/bin/bash
# usage (GPU box): bash tools/sweep_mixed.sh [mixed|f64] -- the 16M sphere on the mixed-record (or fp64) plan over fused depth x rows per partition
set -o pipefail
P=${1:-mixed}
b() { timeout -k 10 250 python bench.py --precision $P --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  %.1f us' % (d['value']/1e9, 1e3*d['roofline']['avg_launch_ms']), d['config']['tiles']['ntiles'], d['config']['tiles']['max_local'])"; }
for kn in "7 975" "8 900" "8 850" "6 1050" "5 1200"; do set -- $kn; echo "$P k=$1 n0=$2  $(FEM_TILE_N0=$2 b --fuse $1)"; done
