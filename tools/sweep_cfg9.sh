#!/bin/bash
# usage (GPU box): bash tools/sweep_cfg9.sh -- 16M sphere on a 1024-thread x 4-row register tile (4096 rows, one workgroup per CU;
# build_ab/libfem_hip_c9.so, FEM_TILE_CONFIG=9) over fused depth x rows per partition, against the default build
set -o pipefail
b() { timeout -k 10 250 python bench.py --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  %.1f us' % (d['value']/1e9, 1e3*d['roofline']['avg_launch_ms']), d['config']['tiles']['ntiles'], d['config']['tiles']['max_local'], d['config']['tiles']['threads'], d['config']['tiles']['rows_per_thread'])"; }
echo "default     $(b)" || exit 1
cp fem_amd/lib/libfem_hip.so /tmp/libfem_hip.keep
cp build_ab/libfem_hip_c9.so fem_amd/lib/libfem_hip.so
for kn in ${SWEEP_KN:-7:2300 7:2420 7:2550 8:2150 8:2256 10:1936 12:1640}; do
  k=${kn%%:*}; n0=${kn##*:}
  echo "cfg9 k=$k n0=$n0  $(FEM_TILE_CONFIG=9 FEM_TILE_N0=$n0 b --fuse $k)" || { cp /tmp/libfem_hip.keep fem_amd/lib/libfem_hip.so; exit 1; }
done
cp /tmp/libfem_hip.keep fem_amd/lib/libfem_hip.so
echo "default     $(b)"
