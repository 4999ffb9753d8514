#!/bin/bash
# usage (GPU box): bash tools/ab_libs_rehearse.sh "TAG ..." [rounds] -- same-box A/B of library builds build_ab/libfem_hip_TAG.so:
# whole mesh (bench.py, 700 steps) and rank 0 of 8 (tools/shard_rehearsal.py, exchange stubbed)
set -o pipefail
cp fem_amd/lib/libfem_hip.so /tmp/libfem_hip.keep
trap 'cp /tmp/libfem_hip.keep fem_amd/lib/libfem_hip.so' EXIT
for round in $(seq 1 ${2:-2}); do
  for tag in $1; do
    cp build_ab/libfem_hip_$tag.so fem_amd/lib/libfem_hip.so
    v=$(timeout -k 10 250 python bench.py --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  %.1f us' % (d['value']/1e9, 1e3*d['roofline']['avg_launch_ms']))") || exit 1
    w=$(timeout -k 10 250 python tools/shard_rehearsal.py sphere16m 8 0 14 700 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.5f ms/step  job %.0f G' % (d['ms_per_step'], d['job_DOF_it_per_s_if_all_ranks_alike']/1e9))") || exit 1
    echo "round $round  $tag  whole $v   rank 0 of 8: $w"
  done
done
