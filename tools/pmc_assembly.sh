#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_assembly.sh TAG [layout]   -- SQ/TCP counter passes of the device assembly (config 5)
set -o pipefail
TAG=$1; LAYOUT=${2:-sell}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- python3 tools/bench_assembly.py 640 20 $LAYOUT > $OUT/sq.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $OUT/tcp -- python3 tools/bench_assembly.py 640 20 $LAYOUT > $OUT/tcp.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_assembly.py 640 20 $LAYOUT > $OUT/stats.log 2>&1 || exit 5
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq", "tcp"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
        for row in csv.DictReader(open(f)):
            if "assemble_rows" in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print("%s %-32s per launch %.4g (launches %d)" % (sub, k, v / max(n, 1), n))
for f in glob.glob("%s/stats/**/*kernel_stats.csv" % out, recursive=True):
    for row in csv.DictReader(open(f)):
        if "assemble" in row["Name"] or "deform" in row["Name"] or "sell_step" in row["Name"]:
            print("stats %-60s calls %s avg_ns %s" % (row["Name"][:60], row["Calls"], row["AverageNs"]))
PY
