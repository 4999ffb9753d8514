set -o pipefail
r() { echo "== $*"; env "$@" timeout -k 10 250 python tools/shard_rehearsal.py sphere16m 8 0 ${KR:-14} 700 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['classes']; print('classes %s ghosts %d  %.5f ms/step  job %.0f G' % (c, d['n_ghost'], d['ms_per_step'], d['job_DOF_it_per_s_if_all_ranks_alike']/1e9))" || return 1; }
timeout -k 10 200 python bench.py --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('single GPU %.1f G DOF-it/s' % (d['value']/1e9))" &&
r X=1 &&
r FEM_OVERLAP=0 &&
r FEM_OVERLAP=0 FEM_TILE_PERSIST=0 &&
KR=28 r FEM_OVERLAP=0 &&
KR=28 r X=1 &&
KR=21 r FEM_OVERLAP=0
