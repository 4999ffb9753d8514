#!/bin/bash
# usage (GPU box): bash tools/r03b_profiles.sh -- kernel traces and HBM counter passes of the round's FINAL kernel build
# (the digest changed with the mixed-record path; summaries: bash tools/r03_summarize.sh r03b)
set -o pipefail
bash tools/profile_round.sh r03b_sphere16m_fused7_rec12 sphere16m || exit 4
bash tools/profile_round.sh r03b_sphere16m_fused7_mixed sphere16m --precision mixed || exit 10
bash tools/profile_round.sh r03b_scan8m_fused5_rec12 scan8m || exit 5
bash tools/profile_round.sh r03b_deform4m deform4m || exit 6
bash tools/profile_round.sh r03b_sphere16m_sell_rec12 sphere16m --fuse 0 || exit 7
bash tools/profile_round.sh r03b_sphere1m_persist sphere1m --steps 448 || exit 9
bash tools/pmc_sq_pass.sh r03b_sphere16m_fused7_rec12_sq sphere16m > gpurun_out/r03b_sphere16m_fused7_rec12_sq_counters.txt 2>&1 || exit 8
echo all done
