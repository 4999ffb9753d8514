#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_sq_imex_fused.sh [TAG] -- two SQ counter passes (<= 8 SQ counters each) of the
# semi-implicit step on the fused tiles at 4.1M vertices: issue mix and LDS behaviour of tile_cheb_kernel and tile_rhs_cheb_kernel
set -o pipefail
TAG=${1:-r03c_imex_fused_sq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq1 -- python3 tools/bench_imex_fused.py 640 4 30 1e-4 > $OUT/sq1.log 2>&1 || exit 3
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq2 -- python3 tools/bench_imex_fused.py 640 4 30 1e-4 > $OUT/sq2.log 2>&1 || exit 4
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for kern in ("tile_cheb_kernel", "tile_rhs_cheb_kernel", "tile_step_kernel"):
    for sub in ("sq1", "sq2"):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
            for row in csv.DictReader(open(f)):
                if "::" + kern + "<" in row["Kernel_Name"]:
                    a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(acc.items()):
            print("%-22s %s %-24s per launch %.4g (launches %d)" % (kern, sub, k, v / max(n, 1), n))
PY
