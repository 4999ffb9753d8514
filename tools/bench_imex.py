#!/usr/bin/env python3
"""Semi-implicit stepping rate: tools/bench_imex.py [nu=640] [steps=20] [cg_iterations=20] [dt factor=10] [tolerance=0] [host]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fem_amd as fa

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 640
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
factor = float(sys.argv[4]) if len(sys.argv) > 4 else 10.0
tol = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0          # > 0: residual-based stop, cg_iterations = upper bound
pts, tri = fa.make_icosphere(nu, 0.1)
part = fa.Mesh(pts, tri).partition(max(1, (10 * nu * nu + 2) // 256))
om = part.ordered
om.init_stats(fa.PREC_F64)
st = om.stats()
model = fa.default_model()
if len(sys.argv) > 6 and sys.argv[6] == "host":     # the reference's CPU constant set (main.cpp:49-54): weak reaction, so dt may be large
    model.r, model.alpha = 1.52, 12.02
dt_exp = 0.1 * st["maxEdgeLen"] ** 2
dt = factor * dt_exp
plan = fa.Plan(om, fa.PREC_F32, implicit=True)
n0, c0 = fa.initial_condition(om.V, 1)
plan.set_state(n0, c0)
if tol > 0:
    plan.imex_tolerance(tol)
plan.run_imex(2, dt, model, iters)
torch.cuda.synchronize()
t0 = time.perf_counter()
plan.run_imex(steps, dt, model, iters)
torch.cuda.synchronize()
el = time.perf_counter() - t0
res = plan.imex_residual()
taken = plan.imex_iterations()
plan.set_state(n0, c0)
plan.run(2, dt_exp, model)
torch.cuda.synchronize()
t0 = time.perf_counter()
plan.run(10 * steps, dt_exp, model)
torch.cuda.synchronize()
ex = (time.perf_counter() - t0) / (10 * steps)
print(json.dumps({"V": om.V, "cg_iterations": iters, "tolerance": tol, "iterations_taken_last_solve": taken, "dt_over_explicit_dt": factor, "ms_per_imex_step": el / steps * 1e3,
                  "ms_per_cg_iteration": el / steps / iters * 1e3, "residual_reduction_n_c": res,
                  "ms_per_explicit_step_sell": ex * 1e3,
                  "simulated_time_per_ms_imex_vs_explicit": (dt / (el / steps)) / (dt_exp / ex),
                  "constants": "host (r 1.52, alpha 12.02)" if model.r < 2 else "kernel (r 59.45, alpha 269)",
                  "state_finite_and_bounded": bool(np.isfinite(plan.get_state()[0]).all() and np.abs(plan.get_state()[0]).max() < 10)}))
