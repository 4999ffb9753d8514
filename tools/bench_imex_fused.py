#!/usr/bin/env python3
"""Semi-implicit stepping on the fused tiles (Chebyshev iteration, fem_plan_run_imex_fused) against the CG path on the
sliced-ELL plan and against the explicit kernels: tools/bench_imex_fused.py [nu=640] [steps=20] [dt factor=30] [tolerance=1e-4] [host|kernel] [scan]
Prints one JSON line; "simulated time per ms" compares how far each scheme advances the model per millisecond.
`scan`: the irregular stand-in of config 4 (valences 3..16, rows of more than 6 entries numbered first, 5 fused steps, 820-row
parts) with the explicit step at 0.2 of the diffusion limit of its WORST row, as bench.py takes it for such meshes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fem_amd as fa

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 640
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
factor = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-4
scan = len(sys.argv) > 6 and sys.argv[6] == "scan"
k = int(os.environ.get("FEM_FUSE_STEPS", "5" if scan else "7"))
pts, tri = fa.make_scanned_like(nu, 0.1, seed=3) if scan else fa.make_icosphere(nu, 0.1)
mesh = fa.Mesh(pts, tri)
if scan:
    mesh, _ = mesh.wide_rows_first(6)
part = mesh.partition(max(1, len(pts) // int(os.environ.get("FEM_TILE_N0", "820" if scan else "975"))))
om = part.ordered
om.init_stats(fa.PREC_F64)
st = om.stats()
model = fa.default_model()
if len(sys.argv) > 5 and sys.argv[5] == "host":
    model.r, model.alpha = 1.52, 12.02
dt_exp = 0.2 / (max(1.0, model.D) * float(np.abs(st["wVtx"]).max())) if scan else 0.1 * st["maxEdgeLen"] ** 2
dt = factor * dt_exp
n0, c0 = fa.initial_condition(om.V, 1)

def timed(fn, reps):
    fn(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(reps); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

tiled = fa.Plan(om, fa.PREC_F32, part=part, ksteps=k)
tiled.set_state(n0, c0)
t_fused = timed(lambda n: tiled.run_imex_fused(n, dt, model, rel_tol=tol), steps)
its = tiled.imex_fused_iterations()
nf, cf = tiled.get_state()
tiled.set_state(n0, c0)
t_exp_fused = timed(lambda n: tiled.run(n, dt_exp, model), 70 * steps)
del tiled
sell = fa.Plan(om, fa.PREC_F32, implicit=True)
sell.set_state(n0, c0)
sell.imex_tolerance(tol)
t_cg = timed(lambda n: sell.run_imex(n, dt, model, 200), steps)
cg_its = sell.imex_iterations()
ncg, ccg = sell.get_state()
sell.set_state(n0, c0)
t_exp_sell = timed(lambda n: sell.run(n, dt_exp, model), 10 * steps)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print(json.dumps({"V": om.V, "mesh": "scanned-like (valences 3..16)" if scan else "icosphere", "spectrum_bound_over_gershgorin": om.spectrum_bound() / (2 * float(np.abs(st["wVtx"]).max())),
                  "dt_over_explicit_dt": factor, "tolerance": tol, "fused_depth": k,
                  "constants": "host (r 1.52, alpha 12.02)" if model.r < 2 else "kernel (r 59.45, alpha 269)",
                  "chebyshev_iterations": its, "ms_per_imex_step_fused_tiles": t_fused * 1e3,
                  "cg_iterations_last_solve": cg_its, "ms_per_imex_step_cg": t_cg * 1e3,
                  "ms_per_explicit_step_single": t_exp_sell * 1e3, "ms_per_explicit_step_fused": t_exp_fused * 1e3,
                  "simulated_time_per_ms_vs_explicit_single_step": {"fused_tiles": factor * t_exp_sell / t_fused, "cg": factor * t_exp_sell / t_cg},
                  "simulated_time_per_ms_vs_explicit_fused": {"fused_tiles": factor * t_exp_fused / t_fused, "cg": factor * t_exp_fused / t_cg},
                  "state_after_%d_steps_rel_l2_fused_vs_cg" % (steps + 2): [rel(nf, ncg), rel(cf, ccg)],
                  "state_finite_and_bounded": bool(np.isfinite(nf).all() and np.abs(nf).max() < 10)}))
