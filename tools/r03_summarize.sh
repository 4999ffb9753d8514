#!/bin/bash
# usage (build machine, after tools/r03_profiles.sh / r03b_profiles.sh has run on the GPU box): bash tools/r03_summarize.sh [r03|r03b]
# condenses gpurun_out/<prefix>_* into profiles/ and records each traffic figure under its run's own traffic key
# (gpurun MERGES into gpurun_out/: delete gpurun_out/<prefix>_* before a repeated profile run, or an older pass's files may be the ones read)
P=${1:-r03}
for tag in ${P}_sphere16m_fused7_rec12 ${P}_sphere16m_fused7_mixed ${P}_scan8m_fused5_rec12 ${P}_deform4m ${P}_sphere16m_sell_rec12 ${P}_sphere1m_persist; do
  d=gpurun_out/$tag
  [ -s $d/bench_under_rocprof.json ] || { echo "missing $d"; continue; }
  key=$(python3 -c "import json,sys; print(json.load(open('$d/bench_under_rocprof.json'))['roofline']['traffic_key'])")
  cp $d/bench_under_rocprof.json profiles/${tag}_bench_under_rocprof.json
  python3 tools/summarize_prof.py $tag $d/stats $d/fetch $d/write "$key" > /dev/null || echo "summary of $tag failed"
  echo "$tag -> $key"
done
cp gpurun_out/${P}_sphere16m_fused7_rec12_sq_counters.txt profiles/ 2>/dev/null
python3 -c "import json; print(json.dumps(json.load(open('profiles/traffic.json')), indent=1)[:3000])"
