#!/bin/bash
# usage (GPU box): bash tools/r03b_profiles.sh [prefix=r03b] -- kernel traces and HBM counter passes of the round's FINAL kernel
# build (every edit of a device source changes the digest that profiles/traffic.json is keyed by: take them again under a new
# prefix; summaries: bash tools/r03_summarize.sh <prefix>)
set -o pipefail
P=${1:-r03b}
bash tools/profile_round.sh ${P}_sphere16m_fused7_rec12 sphere16m || exit 4
bash tools/profile_round.sh ${P}_sphere16m_fused7_mixed sphere16m --precision mixed || exit 10
bash tools/profile_round.sh ${P}_scan8m_fused5_rec12 scan8m || exit 5
bash tools/profile_round.sh ${P}_deform4m deform4m || exit 6
bash tools/profile_round.sh ${P}_sphere16m_sell_rec12 sphere16m --fuse 0 || exit 7
bash tools/profile_round.sh ${P}_sphere1m_persist sphere1m --steps 448 || exit 9
bash tools/pmc_sq_pass.sh ${P}_sphere16m_fused7_rec12_sq sphere16m > gpurun_out/${P}_sphere16m_fused7_rec12_sq_counters.txt 2>&1 || exit 8
echo all done
