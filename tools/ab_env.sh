#!/bin/bash
# usage (GPU box): bash tools/ab_env.sh "VAR=a VAR=b ..." [rounds] [bench args] -- same-box A/B of environment settings on bench.py (700 steps)
set -o pipefail
for round in $(seq 1 ${2:-2}); do
  for setting in $1; do
    v=$(env $setting timeout -k 10 250 python bench.py --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 ${3:-} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  %.1f us' % (d['value']/1e9, 1e3*d['roofline']['avg_launch_ms']))") || exit 1
    echo "round $round  $setting  $v"
  done
done
