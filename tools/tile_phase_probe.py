#!/usr/bin/env python3
"""Diagnostic: time a tiled plan launched with 1, 2, 3, 4 fused steps per launch (same loads,
different amounts of compute) to split the launch time into load and per-step compute."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fem_amd as fa  # noqa: E402

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 640
n0 = int(os.environ.get("FEM_TILE_N0", "620"))
pts, tri = fa.make_icosphere(nu, 0.1)
mesh = fa.Mesh(pts, tri)
part = mesh.partition(mesh.V // n0)
om = part.ordered
om.init_stats(fa.PREC_F64)
dt = 0.1 * om.stats()["maxEdgeLen"] ** 2
plan = fa.Plan(om, fa.PREC_F32, part=part, ksteps=4)
n, c = fa.initial_condition(om.V, 1)
plan.set_state(n, c)
print(plan.tile_info())
model = fa.default_model()
for per in (1, 2, 3, 4, 1, 4):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            plan.run(per, dt, model)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 100
    print("steps/launch %d: %.1f us per launch, %.1f us per step" % (per, el * 1e6, el * 1e6 / per), flush=True)
