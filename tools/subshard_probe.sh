#!/bin/bash
# usage (GPU box): bash tools/subshard_probe.sh [nu] -- one mesh on ONE GPU as 1 plan, and as 2 / 4 in-process ranks (threads,
# each with its own streams, halo by device copies: fem_run --transport inproc): what sub-shards per GPU buy or cost
set -o pipefail
NU=${1:-452}
python3 - "$NU" <<'PY'
import sys; sys.path.insert(0, ".")
import fem_amd as fa
nu = int(sys.argv[1])
pts, tri = fa.make_icosphere(nu, 0.1)
fa.off_write("/tmp/sub_%d.off" % nu, pts, tri)
print("V =", len(pts))
PY
run() { timeout -k 10 300 fem_amd/lib/fem_run /tmp/sub_$NU.off --native --fuse 7 --rows 975 --iterations 350 "$@" 2>&1 | grep -E "Time:|Best time|Ranks:" | tr '\n' ' '; echo; }
echo "one plan:        $(FEM_TILE_PERSIST=0 run)"
echo "one plan (rule): $(run)"
for r in 2 4; do echo "$r ranks inproc:  $(run --ranks $r --transport inproc)"; done
