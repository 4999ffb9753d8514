#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_imex_fused.sh [prefix=r03b] -- rocprofv3 kernel trace + the two HBM counter passes of the
# semi-implicit step on the fused tiles (tools/bench_imex_fused.py 640 10 30 1e-4); raw output under gpurun_out/<prefix>_imex_fused/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r03b}_imex_fused
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_imex_fused.py 640 10 30 1e-4 > $OUT/stats.log 2>&1 || exit 2
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/bench_imex_fused.py 640 4 30 1e-4 > $OUT/fetch.log 2>&1 || exit 3
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/bench_imex_fused.py 640 4 30 1e-4 > $OUT/write.log 2>&1 || exit 4
echo done
