#!/bin/bash
# usage (GPU box): bash tools/sweep_f64_tiles.sh -- fp64 fused plan of the 16.4M sphere over register-tile geometries
# (FEM_TILE_CONFIG: 8 = 512x4 one workgroup per CU, 1 = 256x4 and 3 = 256x3 two per CU, 4 = 256x2 three per CU), fused
# depth and partition size
set -o pipefail
r() { v=$(env "$@" timeout -k 10 250 python bench.py --precision f64 --steps 350 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  %.1f us  tiles %s' % (d['value']/1e9, 1e3*d['roofline']['avg_launch_ms'], d['config']['tiles']['ntiles']))") || return 1; echo "$* : $v"; }
r FEM_TILE_CONFIG=8 FEM_FUSE_STEPS=7 FEM_TILE_N0=975 &&
r FEM_TILE_CONFIG=1 FEM_FUSE_STEPS=5 FEM_TILE_N0=530 &&
r FEM_TILE_CONFIG=1 FEM_FUSE_STEPS=4 FEM_TILE_N0=620 &&
r FEM_TILE_CONFIG=1 FEM_FUSE_STEPS=6 FEM_TILE_N0=450 &&
r FEM_TILE_CONFIG=3 FEM_FUSE_STEPS=4 FEM_TILE_N0=430 &&
r FEM_TILE_CONFIG=4 FEM_FUSE_STEPS=3 FEM_TILE_N0=330 &&
r FEM_TILE_CONFIG=0 FEM_FUSE_STEPS=5 FEM_TILE_N0=530
