"""Semi-implicit (IMEX Euler) stepping on the device.  The reference has no such
path (only unused Jacobian scaffolding, mesh_function.cpp:490-586), so there is nothing of the
reference's to be at parity with: the checker is an independent scipy restatement of the scheme
documented in include/fem_hip.h (sparse direct solves), built on the oracle's assembled operator."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import fem_amd as fa
import oracle_lib as ol
from helpers import parity, scan_case, smooth_ic, sphere_case, stable_dt, strip_case

pytestmark = pytest.mark.gpu


def scipy_imex(st, topo, n, c, dt, steps, model):
    """(I - dt D L) n' = n + dt(-a n Lc - a Gn.Gc + S r n (nMax - n)); (I - dt L) c' = c + dt S (n/(1+n) - c)"""
    nMax, D, r, alpha, S = model
    V = topo.V
    rp, col = topo.rowptr, topo.col
    w = np.asarray(st.wHej)[topo.row_he]
    row = np.repeat(np.arange(V), np.diff(rp))
    L = sp.csr_matrix((w, (row, col)), shape=(V, V)) + sp.diags(np.asarray(st.wVtx))
    I = sp.identity(V, format="csc")
    solve_n = spla.factorized((I - dt * D * L).tocsc())
    solve_c = spla.factorized((I - dt * L).tocsc())
    for _ in range(steps):
        lapc = L @ c
        gn = st.vertex_gradients(st.face_gradients(n))
        gc = st.vertex_gradients(st.face_gradients(c))
        bn = n + dt * (-alpha * n * lapc - alpha * np.einsum("ij,ij->i", gn, gc) + S * r * n * (nMax - n))
        bc = c + dt * S * (n / (1 + n) - c)
        n, c = np.maximum(solve_n(bn), 0), np.maximum(solve_c(bc), 0)
    return n, c


@pytest.mark.parametrize("make,nparts,factor", [(lambda n: sphere_case(16, n), 6, 1.0), (lambda n: sphere_case(16, n), 6, 8.0),
                                                (lambda n: strip_case(n), 3, 4.0), (lambda n: scan_case(16, n), 5, 6.0)])
def test_imex_matches_scipy(make, nparts, factor):
    case = make(nparts)
    case.mesh.init_stats(fa.PREC_F64)
    st = case.ostats(ol.F64)
    dt = factor * stable_dt(st, case.model, safety=1.0)        # factor x the explicit diffusion limit scale
    n0, c0 = smooth_ic(case.pts)
    model = case.model.as_array()
    n_ref, c_ref = scipy_imex(st, case.topo, n0, c0, dt, 5, model)
    plan = fa.Plan(case.mesh, fa.PREC_F64, implicit=True)
    plan.set_state(n0, c0)
    plan.run_imex(5, dt, case.model, cg_iterations=60)
    n, c = plan.get_state()
    assert max(plan.imex_residual()) <= 1e-9
    parity(n, c, n_ref, c_ref, 1e-9)
    # fp32 plan, fewer iterations
    plan32 = fa.Plan(case.mesh, fa.PREC_F32, implicit=True)
    plan32.set_state(n0, c0)
    plan32.run_imex(5, dt, case.model, cg_iterations=25)
    n32, c32 = plan32.get_state()
    parity(n32, c32, n_ref, c_ref, 1e-5)
    # residual-based stop: same answer to the tolerance asked for, in fewer iterations than the bound
    plan.set_state(n0, c0)
    plan.imex_tolerance(1e-10)
    plan.run_imex(5, dt, case.model, cg_iterations=200)
    nt, ct = plan.get_state()
    its = plan.imex_iterations()
    assert 1 <= its < 200 and max(plan.imex_residual()) <= 1e-10, (its, plan.imex_residual())
    parity(nt, ct, n_ref, c_ref, 1e-9)
    plan.imex_tolerance(0.0)
    plan.run_imex(1, dt, case.model, cg_iterations=7)
    assert plan.imex_iterations() == 7


def test_imex_is_stable_where_explicit_is_not():
    """pure diffusion (alpha = r = 0) with a time step 40x the explicit limit: the explicit step
    blows up, the semi-implicit one decays towards the mean and conserves the mass-weighted mean"""
    case = sphere_case(24, 8)
    case.mesh.init_stats(fa.PREC_F64)
    st = case.ostats(ol.F64)
    m = fa.default_model()
    m.alpha, m.r, m.S = 0.0, 0.0, 0.0
    dt = 40 * stable_dt(st, m, safety=1.0)
    rng = np.random.default_rng(2)
    n0 = 1 + 0.01 * rng.standard_normal(case.V)
    c0 = 0.5 + 0.01 * rng.standard_normal(case.V)
    plan = fa.Plan(case.mesh, fa.PREC_F64, implicit=True)
    plan.set_state(n0, c0)
    plan.run_imex(10, dt, m, cg_iterations=80)
    n, c = plan.get_state()
    area = np.asarray(st.areaPerVtx)
    assert np.std(n) < 0.2 * np.std(n0) and np.std(c) < 0.05 * np.std(c0)
    assert abs((area * n).sum() - (area * n0).sum()) <= 1e-10 * area.sum()
    assert abs((area * c).sum() - (area * c0).sum()) <= 1e-10 * area.sum()
    plan.set_state(n0, c0)
    plan.run(10, dt, m)
    ne, _ = plan.get_state()
    assert not np.isfinite(ne).all() or np.std(ne) > 10 * np.std(n0)


def test_imex_requires_flag_and_layout():
    case = sphere_case(6, 2)
    case.mesh.init_stats(fa.PREC_F64)
    with pytest.raises(fa.FemError, match="FEM_PLAN_IMPLICIT"):
        fa.Plan(case.mesh, fa.PREC_F64).run_imex(1, 1e-3)
    with pytest.raises(fa.FemError):
        fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=2, implicit=True)
    with pytest.raises(fa.FemError):
        fa.Plan(case.mesh, fa.PREC_F64, implicit=True).imex_residual()


@pytest.mark.parametrize("make,nparts,factor,ksteps", [(lambda n: sphere_case(16, n), 6, 1.0, 4), (lambda n: sphere_case(16, n), 6, 8.0, 4),
                                                       (lambda n: sphere_case(24, n), 8, 30.0, 7), (lambda n: strip_case(n), 3, 4.0, 3),
                                                       (lambda n: scan_case(16, n), 5, 6.0, 4),
                                                       (lambda n: sphere_case(16, n), 6, 2.0, 1), (lambda n: scan_case(16, n), 5, 2.0, 2)])
def test_imex_on_fused_tiles_matches_scipy(make, nparts, factor, ksteps):
    """fem_plan_run_imex_fused: the same semi-implicit step on a fused-tile plan, the diffusion solves by Chebyshev
    iteration (ksteps iterations per launch, no dot products) -- against the scipy direct solves, in fp64 (tolerance
    1e-12: as close as the CG path), in the reference's precision mix and in fp32 with 12-byte records (1e-6); the
    iteration count is the a-priori one and grows with the time step; the irregular mesh exercises the rows of more than
    6 entries (extras staged in LDS).  From ksteps = 2 on the first launch of a solve is tile_rhs_cheb_kernel (right-hand
    side + ksteps - 1 iterations from the same register rows); ksteps = 1 keeps the right-hand side in a launch of its own"""
    case = make(nparts)
    case.mesh.init_stats(fa.PREC_F64)
    st = case.ostats(ol.F64)
    dt = factor * stable_dt(st, case.model, safety=1.0)
    n0, c0 = smooth_ic(case.pts)
    model = case.model.as_array()
    n_ref, c_ref = scipy_imex(st, case.topo, n0, c0, dt, 5, model)
    plan = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps)
    plan.set_state(n0, c0)
    plan.run_imex_fused(5, dt, case.model, rel_tol=1e-12)
    n, c = plan.get_state()
    its = plan.imex_fused_iterations()
    e = parity(n, c, n_ref, c_ref, 1e-9)
    assert 5 <= its <= 400, its
    plan.run_imex_fused(1, 2 * dt, case.model, rel_tol=1e-12)
    assert plan.imex_fused_iterations() > its
    mixed = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps, mixed=True)
    mixed.set_state(n0, c0)
    mixed.run_imex_fused(5, dt, case.model, rel_tol=1e-12)
    parity(*mixed.get_state(), n_ref, c_ref, 1e-8, tol_dev=1e-5)
    plan32 = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=ksteps)
    plan32.set_state(n0, c0)
    plan32.run_imex_fused(5, dt, case.model, rel_tol=1e-6)
    n32, c32 = plan32.get_state()
    e32 = parity(n32, c32, n_ref, c_ref, 1e-5, tol_dev=1e-3)
    print("fused imex: %d iterations; f64 %s; f32 %s" % (its, e, e32))


def test_imex_on_fused_tiles_solve_within_the_first_launch():
    """a time step so short that the whole solve fits behind the right-hand side in ONE launch (iterations <= ksteps - 1):
    the launch writes the plan's next state itself"""
    case = sphere_case(16, 6)
    case.mesh.init_stats(fa.PREC_F64)
    st = case.ostats(ol.F64)
    dt = 0.02 * stable_dt(st, case.model, safety=1.0)
    n0, c0 = smooth_ic(case.pts)
    n_ref, c_ref = scipy_imex(st, case.topo, n0, c0, dt, 3, case.model.as_array())
    plan = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=8)
    plan.set_state(n0, c0)
    plan.run_imex_fused(3, dt, case.model, rel_tol=1e-9)
    assert 1 <= plan.imex_fused_iterations() <= 7, plan.imex_fused_iterations()
    parity(*plan.get_state(), n_ref, c_ref, 1e-9)


def test_imex_on_fused_tiles_with_negative_weights():
    """a stretched, jittered sheet: most triangles obtuse, a third of the cotangent weights negative, the spectrum of -L
    reaching beyond twice the largest row sum -- the Chebyshev interval must still contain it (per-triangle bound /
    Gershgorin with absolute values, fem_mesh_spectrum_bound), or the iteration diverges instead of matching scipy"""
    from helpers import stretched_case
    case = stretched_case(24, 4)
    case.mesh.init_stats(fa.PREC_F64)
    st = case.ostats(ol.F64)
    w = np.asarray(st.wHej)
    assert (w < 0).mean() > 0.05, (w < 0).mean()
    dt = 20.0 * stable_dt(st, case.model, safety=1.0)
    n0, c0 = smooth_ic(case.pts)
    n_ref, c_ref = scipy_imex(st, case.topo, n0, c0, dt, 4, case.model.as_array())
    plan = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=4)
    plan.set_state(n0, c0)
    plan.run_imex_fused(4, dt, case.model, rel_tol=1e-12)
    parity(*plan.get_state(), n_ref, c_ref, 1e-9)
    assert case.mesh.spectrum_bound() > 2.0 * float(np.abs(np.asarray(st.wVtx)).max())


def test_imex_on_fused_tiles_refuses_what_it_does_not_cover():
    sp_ = sphere_case(6, 2)
    sp_.mesh.init_stats(fa.PREC_F64)
    with pytest.raises(fa.FemError, match="fused-tile"):
        fa.Plan(sp_.mesh, fa.PREC_F64).run_imex_fused(1, 1e-3)
    with pytest.raises(fa.FemError):
        fa.Plan(sp_.mesh, fa.PREC_F64, part=sp_.part, ksteps=2).run_imex_fused(1, 1e-3, rel_tol=0.0)
