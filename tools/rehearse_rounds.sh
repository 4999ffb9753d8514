#!/bin/bash
# usage (GPU box): bash tools/rehearse_rounds.sh -- rank 0 of 8 of the 16.4M sphere (exchange stubbed): partition sizes that put the
# tile count of a launch just under / just over a multiple of the 512 workgroup slots (a launch lasts ceil(tiles / 512) tile
# lifetimes, so 4.4 "rounds" cost 5)
set -o pipefail
r() { echo "== $*"; env "$@" timeout -k 10 250 python tools/shard_rehearsal.py sphere16m 8 0 ${KR:-14} 700 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['classes']; print('classes %s = %d tiles = %.2f rounds, ghosts %d  %.5f ms/step  job %.0f G' % (c, sum(c), sum(c)/512.0, d['n_ghost'], d['ms_per_step'], d['job_DOF_it_per_s_if_all_ranks_alike']/1e9))" || return 1; }
timeout -k 10 200 python bench.py --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('single GPU %.1f G DOF-it/s' % (d['value']/1e9))" &&
r FEM_TILE_N0=975 &&
r FEM_TILE_N0=900 &&
r FEM_TILE_N0=880 &&
r FEM_TILE_N0=865 &&
r FEM_TILE_N0=1040 &&
KR=12 r FEM_FUSE_STEPS=6 FEM_TILE_N0=1050 &&
KR=12 r FEM_FUSE_STEPS=6 FEM_TILE_N0=1090 &&
KR=12 r FEM_FUSE_STEPS=6 FEM_TILE_N0=1110 &&
KR=12 r FEM_FUSE_STEPS=6 FEM_TILE_N0=1130 &&
r FEM_TILE_N0=975
