"""Randomised parity sweep (run by hand on the GPU box: PYTHONPATH=. python tests/fuzz_gpu_parity.py [cases] [seed]):
random mesh kind and size, partition size, fused depth, register tile, step count and shard count; every fp32 / fp64
plan against the f64 oracle, fused against single-step plan.  Not collected by pytest (the fixed cases of
test_gpu_parity.py are the suite); this is for shaking out shapes nobody thought of."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import fem_amd as fa          # noqa: E402
import oracle_lib as ol       # noqa: E402
from helpers import Case, rel_l2, smooth_ic, stable_dt   # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ol.set_threads(16)
worst = {"f32": 0.0, "f64": 0.0, "fused_vs_sell": 0.0, "mixed": 0.0, "imex_cheb_vs_cg": 0.0}
for it in range(cases):
    regular = rng.random() < 0.6
    nu = int(rng.integers(6, 90))
    pts, tri = fa.make_icosphere(nu, 0.1) if regular else fa.make_scanned_like(max(nu, 12), 0.1, seed=int(rng.integers(1, 99)),
                                                                                high_valence_fraction=float(rng.choice([0.001, 0.01, 0.03])))
    V = len(pts)
    ksteps = int(rng.integers(1, 9))
    cfg = int(rng.choice([8, 2, 0]))
    n0 = int(rng.integers(60, 1300))
    nparts = max(1, V // n0)
    steps = int(rng.integers(1, 4 * ksteps + 3))
    os.environ["FEM_TILE_CONFIG"] = str(cfg)
    os.environ["FEM_TILE_PROLOGUE"] = str(int(rng.choice([1, 3])))
    case = Case(pts, tri, nparts, seed=int(rng.integers(1, 1000)))
    case.mesh.init_stats(fa.PREC_F64)
    st = case.ostats(ol.F64)
    dt = case.dt if regular else min(case.dt, stable_dt(st, case.model))
    n0s, c0s = (case.n0, case.c0) if regular else smooth_ic(case.pts)
    n32, c32 = n0s.astype(np.float32).astype(np.float64), c0s.astype(np.float32).astype(np.float64)
    tag = "case %d: %s nu=%d V=%d parts=%d k=%d cfg=%d steps=%d" % (it, "sphere" if regular else "scan", nu, V, nparts, ksteps, cfg, steps)
    try:
        res = {}
        for prec, name, (a, b) in ((fa.PREC_F64, "f64", (n0s, c0s)), (fa.PREC_F32, "f32", (n32, c32))):
            ref = st.step(a, b, dt, steps, case.model.as_array())
            fused = fa.Plan(case.mesh, prec, part=case.part, ksteps=ksteps)
            sell = fa.Plan(case.mesh, prec)
            out = []
            for plan in (fused, sell):
                plan.set_state(a, b)
                plan.run(steps, dt, case.model)
                out.append(plan.get_state())
            e = max(rel_l2(out[0][0], ref[0]), rel_l2(out[0][1], ref[1]), rel_l2(out[1][0], ref[0]), rel_l2(out[1][1], ref[1]))
            worst[name] = max(worst[name], e)
            if name == "f32":
                worst["fused_vs_sell"] = max(worst["fused_vs_sell"], rel_l2(out[0][0], out[1][0]))
            assert e <= (1e-11 if name == "f64" else 1e-5), (name, e)
            assert np.isfinite(out[0][0]).all()
            res[name] = e
        if cfg in (8, 2):
            # the reference's precision mix on the same tiles, against the f64 oracle; and the two semi-implicit solvers
            # (Chebyshev on the fused tiles, conjugate gradients on the sliced-ELL plan) against each other
            ref = st.step(n0s, c0s, dt, steps, case.model.as_array())
            mixed = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps, mixed=True)
            mixed.set_state(n0s, c0s)
            mixed.run(steps, dt, case.model)
            mo = mixed.get_state()
            res["mixed"] = max(rel_l2(mo[0], ref[0]), rel_l2(mo[1], ref[1]))
            worst["mixed"] = max(worst["mixed"], res["mixed"])
            assert res["mixed"] <= 1e-8, ("mixed", res["mixed"])
            tiled = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps)
            cg = fa.Plan(case.mesh, fa.PREC_F64, implicit=True)
            for pl in (tiled, cg):
                pl.set_state(n0s, c0s)
            tiled.run_imex_fused(3, 5 * dt, case.model, rel_tol=1e-12)
            cg.imex_tolerance(1e-12)
            cg.run_imex(3, 5 * dt, case.model, 500)
            (na, ca), (nb, cb) = tiled.get_state(), cg.get_state()
            res["imex"] = max(rel_l2(na, nb), rel_l2(ca, cb))
            worst["imex_cheb_vs_cg"] = max(worst["imex_cheb_vs_cg"], res["imex"])
            assert res["imex"] <= 1e-9, ("imex", res["imex"])
        print("ok  ", tag, "f64 %.1e f32 %.1e packed=%d" % (res["f64"], res["f32"], fused.operator_format()),
              ("mixed %.1e imex %.1e" % (res["mixed"], res["imex"])) if "mixed" in res else "", flush=True)
    except Exception as ex:      # noqa: BLE001
        print("FAIL", tag, type(ex).__name__, ex, flush=True)
        if "FEM_ERR" not in str(ex) and "limit" not in str(ex).lower():
            raise
print("worst", worst)
