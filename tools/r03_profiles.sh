#!/bin/bash
# usage (GPU box): bash tools/r03_profiles.sh -- the round's kernel traces and HBM counter passes in one call
# (summaries: python tools/summarize_prof.py r03_<tag> gpurun_out/r03_<tag>/stats gpurun_out/r03_<tag>/fetch gpurun_out/r03_<tag>/write KEY,
#  KEY = roofline.traffic_key of gpurun_out/r03_<tag>/bench_under_rocprof.json)
set -o pipefail
bash tools/profile_round.sh r03_sphere16m_fused7_rec12 sphere16m || exit 4
bash tools/profile_round.sh r03_scan8m_fused5_rec12 scan8m || exit 5
bash tools/profile_round.sh r03_deform4m deform4m || exit 6
bash tools/profile_round.sh r03_sphere16m_sell_rec12 sphere16m --fuse 0 || exit 7
bash tools/profile_round.sh r03_sphere1m_persist sphere1m --steps 448 || exit 9
bash tools/pmc_sq_pass.sh r03_sphere16m_fused7_rec12_sq sphere16m > gpurun_out/r03_sphere16m_fused7_rec12_sq_counters.txt 2>&1 || exit 8
echo all done
