#!/bin/bash
# usage (GPU box, repo root): bash tools/run_all_configs.sh OUT.jsonl
# The five BASELINE.md configurations on one GPU, one JSON line each (appended as they finish).  Every bench.py line
# verifies itself against the oracle (runs longer than 64 steps: 25 steps on the same plan before the timed region).
OUT=${1:-gpurun_out/configs.jsonl}
: > $OUT
run() { echo "== $*" >&2; PYTHONPATH=. timeout -k 10 500 "$@" 2>/dev/null | tail -1 >> $OUT || echo "{\"failed\": \"$*\"}" >> $OUT; tail -c 300 $OUT >&2; echo >&2; }
run python bench.py --workload sphere10k --steps 2000 --no-cpu-baseline --no-variants            # config 1, fused
run python bench.py --workload sphere10k --steps 2000 --fuse 0 --no-cpu-baseline   # config 1, single-step kernel
run python tools/bench_ref_step.py 32 197 2000                                     # config 1, drop-in step()
run python bench.py --workload sphere1m --steps 1000 --no-cpu-baseline --no-variants             # config 2
run python bench.py --workload sphere1m --steps 1000 --fuse 0 --no-cpu-baseline
run python bench.py --workload sphere16m --steps 1000                              # config 3 (headline, with cpu_baseline and the fp64 / four-float variants)
run python bench.py --workload sphere16m --steps 400 --fuse 0 --no-cpu-baseline
run python bench.py --workload scan8m --steps 400 --no-cpu-baseline                # config 4
run python bench.py --workload scan8m --steps 200 --fuse 0 --no-cpu-baseline
run python bench.py --workload deform4m --steps 200 --no-cpu-baseline              # config 5 as the loop (fused assembly-and-step)
run python bench.py --gpus 2 --dist-backend gloo --workload sphere1m --steps 20 --warmup 5       # the N > 1 line (rehearsal: two ranks on this GPU, halo through the host)
run python bench.py --gpus 2 --dist-backend gloo --workload deform4m --steps 20 --warmup 5      # config 5 on shards (rehearsal)
run python tools/bench_assembly.py 640 200 sell                                    # config 5, the kernels apart
run python tools/bench_assembly.py 640 200 tile
run python tools/bench_ref_step.py 640 250 100                                     # drop-in step() at 4M
run python tools/bench_imex.py 640 20 30 30 1e-4                                   # semi-implicit stepping, kernel constants, dt = 30 x explicit
run python tools/bench_imex.py 640 20 60 160 1e-4 host                             # the same with the reference's host constants, dt = 160 x explicit
run python tools/bench_imex_fused.py 640 20 30 1e-4                                # semi-implicit stepping on the fused tiles (Chebyshev), kernel constants
run python tools/bench_imex_fused.py 640 20 160 1e-4 host                          # the same with the reference's host constants
run python bench.py --workload sphere16m --steps 1000 --precision mixed --no-cpu-baseline   # config 3 in the reference's own precision mix
run python tools/bench_imex_fused.py 900 10 30 1e-4 kernel scan                   # semi-implicit stepping on the irregular 8.1M mesh of config 4, dt = 30 x its explicit step
run python tools/bench_imex_fused.py 900 10 200 1e-4 kernel scan                  # ... dt = 200 x
