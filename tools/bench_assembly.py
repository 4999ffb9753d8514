#!/usr/bin/env python3
"""BASELINE config 5: deforming mesh, re-assembly every step (assemble + step per iteration).
Reports elements/s and algorithmic GB/s (68 B per element) for the assembly, and the
combined iteration rate.  usage: tools/bench_assembly.py [nu] [iters] [layout: sell|tile]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fem_amd as fa  # noqa: E402

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 640
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
layout = sys.argv[3] if len(sys.argv) > 3 else "sell"
pts, tri = fa.make_icosphere(nu, 0.1)
mesh = fa.Mesh(pts, tri)
part = mesh.partition(mesh.V // (620 if layout == "tile" else 256))
om = part.ordered
om.init_stats(fa.PREC_F64)
dt = 0.1 * om.stats()["maxEdgeLen"] ** 2
plan = fa.Plan(om, fa.PREC_F32, part=part, ksteps=1 if layout == "tile" else 0, assemble=True)
n, c = fa.initial_condition(om.V, 1)
plan.set_state(n, c)
model = fa.default_model()
q = 2 * 3.141592653589793 / (40 * 0.1)


def timed(fn, reps):
    fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t_asm = timed(lambda i: plan.assemble(), iters)
t_def = timed(lambda i: plan.deform(0.01, q, 0.01 * i), iters)
t_step = timed(lambda i: plan.run(1, dt, model), iters)


def iteration(i):
    plan.deform(0.01, q, 0.01 * i)
    plan.assemble()
    plan.run(1, dt, model)


t_iter = timed(iteration, iters)
F = om.F
print(json.dumps({"workload": "deforming icosphere nu=%d, V=%d, F=%d, layout=%s, fp32" % (nu, om.V, F, layout),
                  "assemble_ms": t_asm * 1e3, "deform_ms": t_def * 1e3, "step_ms": t_step * 1e3,
                  "iteration_ms": t_iter * 1e3, "elements_per_s": F / t_asm,
                  "assembly_alg_GBs": 68.0 * F / t_asm / 1e9, "assembly_frac_of_8TBs": 68.0 * F / t_asm / 8e12,
                  "iteration_DOF_it_per_s": 2.0 * om.V / t_iter}))
