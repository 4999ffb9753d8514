#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_round.sh TAG WORKLOAD [extra bench args]
# rocprofv3 kernel trace + the two HBM traffic passes (FETCH_SIZE, WRITE_SIZE: separate --pmc runs)
# of `python3 bench.py --workload WORKLOAD`; raw output under gpurun_out/, summaries via
# tools/summarize_prof.py into profiles/ afterwards (on the build machine).
set -o pipefail
TAG=$1; WL=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --workload $WL --steps 200 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 "$@" > $OUT/stats.log 2>&1 || exit 2
grep '^{"metric"' $OUT/stats.log > $OUT/bench_under_rocprof.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --workload $WL --steps 40 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 "$@" > $OUT/fetch.log 2>&1 || exit 3
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --workload $WL --steps 40 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 "$@" > $OUT/write.log 2>&1 || exit 4
echo done $TAG
