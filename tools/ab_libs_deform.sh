#!/bin/bash
# usage (GPU box): bash tools/ab_libs_deform.sh "TAG ..." [rounds] -- same-box A/B of library builds on the config-5 loop (deform4m)
set -o pipefail
cp fem_amd/lib/libfem_hip.so /tmp/libfem_hip.keep
trap 'cp /tmp/libfem_hip.keep fem_amd/lib/libfem_hip.so' EXIT
for round in $(seq 1 ${2:-2}); do
  for tag in $1; do
    cp build_ab/libfem_hip_$tag.so fem_amd/lib/libfem_hip.so
    d=$(timeout -k 10 250 python bench.py --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 --workload deform4m --steps 400 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  kernel %.4f ms' % (d['value']/1e9, d['roofline']['avg_launch_ms']))") || exit 1
    echo "round $round  $tag  deform4m $d"
  done
done
