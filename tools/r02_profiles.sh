#!/bin/bash
# usage (GPU box): bash tools/r02_profiles.sh -- the round's judged numbers in one call
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err || exit 2
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 --precision f64 > gpurun_out/r02_bench_f64.json 2> gpurun_out/r02_bench_f64.err || exit 3
bash tools/profile_round.sh r02_sphere16m_fused7 sphere16m || exit 4
bash tools/profile_round.sh r02_deform4m deform4m || exit 5
echo all done
