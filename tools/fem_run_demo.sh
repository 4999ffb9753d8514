#!/bin/bash
# fem_run on config 1 (reference protocol) and on the 1M sphere (native plan), from .off files
set -e
mkdir -p gpurun_out
python - <<'PY'
import fem_amd as fa
for nu, name in ((32, "sphere32"), (320, "sphere320")):
    p, t = fa.make_icosphere(nu, 0.1)
    fa.off_write("gpurun_out/%s.off" % name, p, t)
PY
{
echo "\$ fem_run sphere32.off -v"
timeout -k 10 120 fem_amd/lib/fem_run gpurun_out/sphere32.off -v
echo "\$ fem_run sphere32.off"
timeout -k 10 120 fem_amd/lib/fem_run gpurun_out/sphere32.off
echo "\$ fem_run sphere32.off --native -v"
timeout -k 10 120 fem_amd/lib/fem_run gpurun_out/sphere32.off --native -v
echo "\$ fem_run sphere320.off -n 4096 -v"
timeout -k 10 200 fem_amd/lib/fem_run gpurun_out/sphere320.off -n 4096 -v
echo "\$ fem_run sphere320.off --native -v --out pattern.off"
timeout -k 10 200 fem_amd/lib/fem_run gpurun_out/sphere320.off --native -v --out gpurun_out/pattern320.off
} > gpurun_out/fem_run_demo.txt 2>&1
rm -f gpurun_out/sphere32.off gpurun_out/sphere320.off gpurun_out/pattern320.off
cat gpurun_out/fem_run_demo.txt
