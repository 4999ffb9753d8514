#!/bin/bash
# usage (GPU box): bash tools/ab_libs_ref.sh "TAG ..." [rounds] -- same-box A/B of library builds on the drop-in step() (4.1M sphere and config 1)
set -o pipefail
cp fem_amd/lib/libfem_hip.so /tmp/libfem_hip.keep
trap 'cp /tmp/libfem_hip.keep fem_amd/lib/libfem_hip.so' EXIT
for round in $(seq 1 ${2:-2}); do
  for tag in $1; do
    cp build_ab/libfem_hip_$tag.so fem_amd/lib/libfem_hip.so
    e=$(PYTHONPATH=. timeout -k 10 200 python tools/bench_ref_step.py 640 250 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.2f G  %.4f ms' % (d['DOF_it_per_s']/1e9, d['ms_per_step']))") || exit 1
    f=$(PYTHONPATH=. timeout -k 10 200 python tools/bench_ref_step.py 32 197 2000 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.2f G' % (d['DOF_it_per_s']/1e9))") || exit 1
    echo "round $round  $tag  step() 4.1M: $e   10k: $f"
  done
done
