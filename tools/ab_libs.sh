#!/bin/bash
# usage (GPU box): bash tools/ab_libs.sh "TAG ..." [rounds] -- same-box A/B of library builds build_ab/libfem_hip_TAG.so:
# each is copied over fem_amd/lib/libfem_hip.so in turn and bench.py (16M sphere, 700 steps) is run; `rounds` passes
set -o pipefail
cp fem_amd/lib/libfem_hip.so /tmp/libfem_hip.keep
for round in $(seq 1 ${2:-2}); do
  for tag in $1; do
    cp build_ab/libfem_hip_$tag.so fem_amd/lib/libfem_hip.so
    v=$(timeout -k 10 250 python bench.py --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 ${AB_ARGS:-} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  %.1f us' % (d['value']/1e9, 1e3*d['roofline']['avg_launch_ms']))") || { cp /tmp/libfem_hip.keep fem_amd/lib/libfem_hip.so; exit 1; }
    echo "round $round  $tag  $v"
  done
done
cp /tmp/libfem_hip.keep fem_amd/lib/libfem_hip.so
