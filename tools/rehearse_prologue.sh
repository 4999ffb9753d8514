#!/bin/bash
# usage (GPU box): bash tools/rehearse_prologue.sh -- rank 0 of 8 of the 16.4M sphere (exchange stubbed) with the operator streams cached
# (FEM_TILE_PROLOGUE=1) or nontemporal (3): a 2M-row shard's operator (~213 MB) plus its state is about the size of the Infinity Cache
set -o pipefail
r() { echo "== $*"; env "$@" timeout -k 10 250 python tools/shard_rehearsal.py sphere16m 8 0 ${KR:-14} 1400 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['classes']; print('classes %s ghosts %d  %.5f ms/step  job %.0f G' % (c, d['n_ghost'], d['ms_per_step'], d['job_DOF_it_per_s_if_all_ranks_alike']/1e9))" || return 1; }
r X=1 &&
r FEM_TILE_PROLOGUE=3 &&
r FEM_TILE_PROLOGUE=1 &&
r FEM_TILE_PROLOGUE=2 &&
r FEM_TILE_PROLOGUE=3 &&
r X=1
