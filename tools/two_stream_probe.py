#!/usr/bin/env python3
"""usage (GPU box): python tools/two_stream_probe.py [nu] [steps] -- can the tail of one fused launch hide behind the
head of another?  Two INDEPENDENT plans of the same nu-sphere (no data in common) take `steps` time steps each:
 (a) one after the other's launches on ONE stream (alternating launch by launch),
 (b) each on a stream of its own, concurrently (the second one started half a launch late).
Prints the aggregate rate of both.  What it decides: whether a two-colour schedule of ONE plan's tiles (two halves on
two streams, interface tiles first) could take the per-launch fixed cost (~30 us) off shard-sized launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fem_amd as fa
from fem_amd import dist as fdist

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 452          # 2.04M rows: the size of one of eight shards
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 700
k = 7
os.environ.setdefault("FEM_TILE_PERSIST", "0")
os.environ.setdefault("FEM_TILE_CONFIG", "8")
pts, tri = fa.make_icosphere(nu, 0.1)
runs = [fdist.SingleRun(fa.Mesh(pts, tri), len(pts) // 975, "f32", ksteps=k) for _ in range(2)]
V = runs[0].V
print("V = %d, %d tiles per plan, groups_per_launch %s" % (V, runs[0].plan.tile_info()["ntiles"], runs[0].plan.tile_info().get("groups_per_launch")))
s = [torch.cuda.Stream(), torch.cuda.Stream()]
launches = steps // k

def one_stream():
    st = s[0]
    for _ in range(launches):
        for r in runs:
            r.plan.run(k, r.dt, r.model, st.cuda_stream)

def two_streams():
    runs[0].plan.run(k, runs[0].dt, runs[0].model, s[0].cuda_stream)
    for i in range(launches - 1):
        runs[1].plan.run(k, runs[1].dt, runs[1].model, s[1].cuda_stream)
        runs[0].plan.run(k, runs[0].dt, runs[0].model, s[0].cuda_stream)
    runs[1].plan.run(k, runs[1].dt, runs[1].model, s[1].cuda_stream)

def single():
    for _ in range(launches):
        runs[0].plan.run(k, runs[0].dt, runs[0].model, s[0].cuda_stream)

for name, fn, nplans in (("single plan, one stream", single, 1), ("two plans, one stream", one_stream, 2), ("two plans, two streams", two_streams, 2)) * 2:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("%-26s %.1f G DOF-it/s aggregate, %.1f us per launch" % (name, 2.0 * V * launches * k * nplans / el / 1e9, el / (launches * nplans) * 1e6))
