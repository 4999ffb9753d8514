#!/bin/bash
# usage (GPU box): bash tools/ab_libs_multi.sh "TAG ..." [rounds] -- same-box A/B of library builds build_ab/libfem_hip_TAG.so on
# the headline (sphere16m fused), the single-step kernel, the deforming-mesh loop and the drop-in step
set -o pipefail
cp fem_amd/lib/libfem_hip.so /tmp/libfem_hip.keep
trap 'cp /tmp/libfem_hip.keep fem_amd/lib/libfem_hip.so' EXIT
b() { timeout -k 10 250 python bench.py --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G' % (d['value']/1e9))"; }
for round in $(seq 1 ${2:-2}); do
  for tag in $1; do
    cp build_ab/libfem_hip_$tag.so fem_amd/lib/libfem_hip.so
    a=$(b --steps 700 --warmup 7) || exit 1
    c=$(b --steps 300 --warmup 5 --fuse 0) || exit 1
    d=$(b --workload deform4m --steps 300 --warmup 10) || exit 1
    e=$(PYTHONPATH=. timeout -k 10 200 python tools/bench_ref_step.py 640 250 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.2f G' % (d['DOF_it_per_s']/1e9))") || exit 1
    echo "round $round  $tag  fused $a  sell $c  deform4m $d  ref_step $e"
  done
done
