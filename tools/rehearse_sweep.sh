#!/bin/bash
# usage (GPU box): bash tools/rehearse_sweep.sh  -- one of eight ranks of the 16.4M sphere on one GPU (exchange stubbed), variants
set -o pipefail
r() { echo "== $*"; env "$@" timeout -k 10 250 python tools/shard_rehearsal.py sphere16m 8 0 ${KR:-14} 700 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('classes %s ghosts %d  %.5f ms/step  job %.0f G' % (d['classes'], d['n_ghost'], d['ms_per_step'], d['job_DOF_it_per_s_if_all_ranks_alike']/1e9))" || return 1; }
timeout -k 10 200 python bench.py --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('single GPU %.1f G DOF-it/s' % (d['value']/1e9))" &&
r FEM_BOUNDARY_STREAM=0 &&
r FEM_BOUNDARY_STREAM=1 &&
KR=7 r FEM_BOUNDARY_STREAM=1 &&
KR=7 r FEM_BOUNDARY_STREAM=0 &&
r FEM_BOUNDARY_STREAM=1 FEM_TILE_N0=1000 &&
r FEM_BOUNDARY_STREAM=1 FEM_TILE_N0=800 &&
KR=12 r FEM_BOUNDARY_STREAM=1 FEM_TILE_CONFIG=2 FEM_FUSE_STEPS=6 FEM_TILE_N0=760 &&
r FEM_OVERLAP=0
