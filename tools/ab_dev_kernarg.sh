#!/bin/bash
b() { timeout -k 10 250 python bench.py --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G' % (d['value']/1e9))"; }
# usage (GPU box): bash tools/ab_dev_kernarg.sh -- bench.py with and without HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory) on three workloads, two rounds
for r in 1 2; do
for wl in "sphere10k --steps 4000" "sphere1m --steps 1000" "sphere16m --steps 20 --warmup 5"; do
  echo "round $r $wl: default $(b --workload $wl)   dev-kernarg $(HIP_FORCE_DEV_KERNARG=1 b --workload $wl)"
done; done
