#!/usr/bin/env python3
"""EXPERIMENT (round 3): tile_step_persist_kernel (FEM_TILE_PERSIST = groups of k steps per launch, tiles handed over
through per-tile flags inside the launch) against the launch-per-group schedule: bit-for-bit comparison of the two
(both kernels are deterministic, so ONE stale row shows) and their rates.
usage: tools/persist_probe.py [nu=320] [steps=70] [groups per launch=10] [deps=1] [time steps=0]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fem_amd as fa

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 320
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 70
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 10
deps = sys.argv[4] if len(sys.argv) > 4 else "1"
tsteps = int(sys.argv[5]) if len(sys.argv) > 5 else 0
k = 7
pts, tri = fa.make_icosphere(nu, 0.1)
mesh = fa.Mesh(pts, tri)
part = mesh.partition(max(1, mesh.V // 975))
om = part.ordered
om.init_stats(fa.PREC_F64)
dt = 0.1 * om.stats()["maxEdgeLen"] ** 2
model = fa.default_model()
n0, c0 = fa.initial_condition(om.V, 1)
os.environ["FEM_TILE_CONFIG"] = "8"
plain = fa.Plan(om, fa.PREC_F32, part=part, ksteps=k)
os.environ["FEM_TILE_PERSIST"] = str(groups)
os.environ["FEM_TILE_PERSIST_DEPS"] = deps
pers = fa.Plan(om, fa.PREC_F32, part=part, ksteps=k)
del os.environ["FEM_TILE_PERSIST"]
out = {"V": om.V, "tiles": plain.tile_info()["ntiles"], "steps": steps, "groups_per_launch": groups, "deps": deps}
res = []
print("plans built", out, file=sys.stderr, flush=True)
for name, plan in (("plain", plain), ("persist", pers)):
    plan.set_state(n0, c0)
    plan.run(steps, dt, model)
    print(name, "queued", file=sys.stderr, flush=True)
    res.append(plan.get_state())
    print(name, "done", file=sys.stderr, flush=True)
out["bit_identical"] = bool(np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]))
out["rows_differing"] = int((res[0][0] != res[1][0]).sum())
out["max_abs_diff"] = float(np.abs(res[0][0] - res[1][0]).max())
if tsteps > 0:
    # alternating rounds in one process: the two schedules see the same clocks
    rounds = {"plain": [], "persist": []}
    for rnd in range(3):
        for name, plan in (("plain", plain), ("persist", pers)):
            plan.run(2 * k, dt, model)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.run(tsteps, dt, model)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            rounds[name].append(round(2.0 * om.V * tsteps / el / 1e9, 1))
    for name in rounds:
        out[name + "_G_DOF_it_per_s"] = sorted(rounds[name])[1]
        out[name + "_rounds"] = rounds[name]
    plain.get_state(); pers.get_state()      # raises if the persistent kernel gave up
print(json.dumps(out))
