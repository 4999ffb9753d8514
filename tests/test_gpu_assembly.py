"""Device assembly (initMeshStats on the GPU, reference src/mesh_function.cpp:27-397) against the
CPU oracle: per-face records (A1, A2, A4, A5) and the assembled operator (A3 + folded gradients),
then time steps on a device-assembled, deformed mesh.
Bars: fp64 plan <= 1e-11 relative; fp32 plan (the reference's Vec3f geometry) <= 2e-5 relative
against the mixed-precision oracle -- the reference's own float geometry has that round-off."""
import numpy as np
import pytest

import fem_amd as fa
import oracle_lib as ol
from helpers import Case, parity, rel_l2, scan_case, sphere_case, strip_case

pytestmark = pytest.mark.gpu


def check_records(rec, st, F, tol):
    he = st.meshAngle[:3 * F].reshape(F, 3)          # half-edge 3f+k ends at corner (k+1)%3
    cot = st.meshCot[:3 * F].reshape(F, 3)
    assert np.abs(rec[:, 0:3] - he[:, [2, 0, 1]]).max() <= tol * np.pi
    assert np.abs(rec[:, 4:7] - cot[:, [2, 0, 1]]).max() <= 50 * tol * np.abs(cot).max()
    assert np.abs(rec[:, 3] - st.meshArea).max() <= tol * st.meshArea.max()
    g12, g13 = np.asarray(st.gradVec12, np.float64), np.asarray(st.gradVec13, np.float64)
    assert np.abs(rec[:, 8:11] - g12).max() <= tol * np.abs(g12).max()
    assert np.abs(rec[:, 12:15] - g13).max() <= tol * np.abs(g13).max()
    assert abs(rec[:, 7].max() - st.maxEdgeLen) <= tol * st.maxEdgeLen


@pytest.mark.parametrize("make,nparts", [(lambda n: sphere_case(12, n), 6), (lambda n: strip_case(n), 3),
                                         (lambda n: scan_case(16, n), 5)])
@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("ksteps", [0, 4])
def test_device_assembly_matches_oracle(make, nparts, prec, ksteps):
    case = make(nparts)
    case.mesh.init_stats(fa.PREC_F64 if prec == "f64" else fa.PREC_MIXED)
    plan = fa.Plan(case.mesh, fa.PREC_F64 if prec == "f64" else fa.PREC_F32, part=case.part, ksteps=ksteps,
                   assemble=True)
    host_coef = plan.coef()                           # packed from the host-assembled MeshStats
    plan.assemble()
    tol = 1e-11 if prec == "f64" else 2e-5
    st = case.ostats(ol.F64 if prec == "f64" else ol.MIXED)
    check_records(plan.face_records(), st, case.F, tol)
    assert abs(plan.max_edge() - st.maxEdgeLen) <= tol * st.maxEdgeLen
    dev_coef = plan.coef()
    assert dev_coef.shape == host_coef.shape
    scale = np.abs(host_coef).max(axis=0)
    assert (np.abs(dev_coef - host_coef).max(axis=0) <= 20 * tol * scale).all()


def deformed(pts, amp, q, phase):
    s = 1 + amp * np.sin(q * pts[:, 0] + phase) * np.cos(q * pts[:, 1] - 0.5 * phase) * np.sin(q * pts[:, 2] + 0.25 * phase)
    return pts * s[:, None]


@pytest.mark.parametrize("ksteps", [0, 3])
def test_steps_on_device_assembled_deformed_mesh(ksteps):
    """BASELINE config 5 shape: deform -> re-assemble on the device -> step, against the oracle
    assembling the same deformed positions on the CPU"""
    case = sphere_case(16, 8)
    case.mesh.init_stats(fa.PREC_F64)
    plan = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps, assemble=True)
    amp, q, phase = 0.03, 0.9, 0.7
    plan.deform(amp, q, phase)
    plan.assemble()
    pts_d = deformed(case.pts, amp, q, phase)
    st = ol.Stats(case.topo, pts_d, ol.F64)
    assert abs(plan.max_edge() - st.maxEdgeLen) <= 1e-12 * st.maxEdgeLen
    check_records(plan.face_records(), st, case.F, 1e-11)
    dt = 0.1 * st.maxEdgeLen ** 2
    plan.set_state(case.n0, case.c0)
    plan.run(6, dt, case.model)
    n_gpu, c_gpu = plan.get_state()
    n_ref, c_ref = st.step(case.n0, case.c0, dt, 6, case.model.as_array())
    parity(n_gpu, c_gpu, n_ref, c_ref, 1e-11)
    # set_points: new base positions from the host (ordered numbering)
    plan.set_points(case.pts)
    plan.assemble()
    check_records(plan.face_records(), case.ostats(ol.F64), case.F, 1e-11)


def test_assembly_requires_flag():
    case = sphere_case(4, 2)
    case.mesh.init_stats(fa.PREC_F64)
    plan = fa.Plan(case.mesh, fa.PREC_F64)
    with pytest.raises(fa.FemError):
        plan.assemble()


@pytest.mark.parametrize("make,nparts", [(lambda n: sphere_case(12, n), 6), (lambda n: strip_case(n), 3),
                                         (lambda n: scan_case(16, n), 5)])
@pytest.mark.parametrize("ksteps", [0, 4])
def test_fan_and_slot_table_assembly_agree(make, nparts, ksteps, monkeypatch):
    """the fan-ordered fast path (each face once per row) and the slot-table kernel (both faces of every
    edge) form the same terms in the same order of addition; the compiler contracts them into FMAs
    differently, so they agree to a few ulp of each column's scale, not bit for bit"""
    case = make(nparts)
    case.mesh.init_stats(fa.PREC_MIXED)
    coefs = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("FEM_ASSEMBLE_GENERIC", "1")
        plan = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=ksteps, assemble=True)
        plan.assemble()
        coefs.append((plan.coef().copy(), plan.max_edge()))
    scale = np.abs(coefs[0][0]).max(axis=0)
    assert (np.abs(coefs[0][0] - coefs[1][0]).max(axis=0) <= 4e-7 * scale).all() and coefs[0][1] == coefs[1][1]


def test_non_manifold_vertex_falls_back_to_slot_table():
    """two tetrahedra that share one vertex: its row is two fans, so the plan cannot use the fan kernel;
    the operator still matches the host assembly"""
    tet = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
    pts = np.vstack([tet, -tet[1:] * 0.8])
    faces = [(0, 2, 1), (0, 1, 3), (0, 3, 2), (1, 2, 3)]
    remap = {0: 0, 1: 4, 2: 5, 3: 6}
    tri = np.array(faces + [tuple(remap[v] for v in f) for f in faces], np.uint32)
    mesh = fa.Mesh(pts, tri)
    part = mesh.partition(1)
    om = part.ordered
    om.init_stats(fa.PREC_F64)
    for ksteps in (0, 2):
        plan = fa.Plan(om, fa.PREC_F64, part=part, ksteps=ksteps, assemble=True)
        host = plan.coef().copy()
        plan.assemble()
        dev = plan.coef()
        assert np.abs(dev - host).max() <= 1e-11 * np.abs(host).max()


def _config5_loop(case, prec, mode, iters=6):
    """deform(phase_i) -> assemble -> one step, `iters` times; returns the final state and the per-iteration
    maxEdgeLen the device reported.  mode: "sell" / "tile" = assembly kernel then step kernel on that layout,
    "fused" / "fused+store" = fem_plan_assemble_step"""
    case.mesh.init_stats(fa.PREC_F64 if prec == "f64" else fa.PREC_MIXED)
    plan = fa.Plan(case.mesh, fa.PREC_F64 if prec == "f64" else fa.PREC_F32, part=case.part,
                   ksteps=2 if mode == "tile" else 0, assemble=True)
    plan.set_state(case.n0, case.c0)
    edges = []
    for i in range(iters):
        plan.deform(0.03, 0.9, 0.35 * i)
        if mode.startswith("fused"):
            plan.assemble_step(case.dt_loop, case.model, store_operator=mode.endswith("store"))
        else:
            plan.assemble()
            plan.run(1, case.dt_loop, case.model)
        edges.append(plan.max_edge())
    return plan.get_state(), edges, plan


def _config5_oracle(case, prec, iters=6):
    n, c = case.n0.copy(), case.c0.copy()
    if prec == "f32":
        n, c = n.astype(np.float32).astype(np.float64), c.astype(np.float32).astype(np.float64)
    edges = []
    for i in range(iters):
        st = ol.Stats(case.topo, deformed(case.pts, 0.03, 0.9, 0.35 * i), ol.F64 if prec == "f64" else ol.MIXED)
        n, c = st.step(n, c, case.dt_loop, 1, case.model.as_array())
        edges.append(st.maxEdgeLen)
    return (n, c), edges


@pytest.mark.parametrize("make,nparts", [(lambda n: sphere_case(20, n), 8),
                                         (lambda n: Case(*fa.make_scanned_like(20, 0.1, seed=5, flip_fraction=0.15, high_valence_fraction=0.0), n), 6),
                                         (lambda n: scan_case(16, n), 5)])
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_config5_iteration_loop_vs_oracle(make, nparts, prec):
    """BASELINE config 5 as a loop: six iterations of deform -> re-assemble -> step with a different deformation
    each time, against the oracle re-assembling the deformed positions on the CPU every iteration.  All four
    device paths: assembly + step kernels on the sliced-ELL and the tile layout, and the fused kernel with and
    without storing the operator (regular mesh: 6-entry rows; flipped mesh: up to 12; fanned mesh: wider rows,
    where fem_plan_assemble_step falls back to the two kernels)."""
    case = make(nparts)
    st0 = case.ostats(ol.F64)
    case.dt_loop = min(0.08 * st0.maxEdgeLen ** 2, 0.8 * 0.2 / (max(1.0, case.model.D) * np.abs(st0.wVtx).max()))
    (n_ref, c_ref), e_ref = _config5_oracle(case, prec)
    tol = 1e-11 if prec == "f64" else 2e-5
    # of the deviation from the homogeneous state: fp64 1e-9; fp32 1e-4 -- both sides build the operator from fp32
    # geometry but by different (equivalent) formulas (cot = (|e|^2 - e.e')/|c| here, 1/tan(acos(.)) in the reference's
    # order of operations there), so the weights differ in their last fp32 digits: observed 2.0e-5 after six iterations
    tol_dev = None if prec == "f64" else 1e-4
    outs = {}
    for mode in ("sell", "tile", "fused", "fused+store"):
        (n, c), edges, plan = _config5_loop(case, prec, mode)
        parity(n, c, n_ref, c_ref, tol, tol_dev=tol_dev, what=mode)
        assert np.abs(np.array(edges) - np.array(e_ref)).max() <= (1e-12 if prec == "f64" else 2e-5) * max(e_ref), mode
        outs[mode] = (n, c)
        if mode == "fused+store":          # the stored operator is the one of the last deformation: one more plain step on it
            plan.run(1, case.dt_loop, case.model)
            st = ol.Stats(case.topo, deformed(case.pts, 0.03, 0.9, 0.35 * 5), ol.F64 if prec == "f64" else ol.MIXED)
            n2, c2 = st.step(n_ref, c_ref, case.dt_loop, 1, case.model.as_array())
            g = plan.get_state()
            parity(g[0], g[1], n2, c2, tol, tol_dev=tol_dev)
    # fused and unfused apply the same coefficients in the same order (two kernels: the compiler may contract a
    # multiply-add differently, hence a few ulp, not bit for bit)
    for other in ("sell", "fused+store"):
        assert rel_l2(outs["fused"][0], outs[other][0]) <= (1e-13 if prec == "f64" else 1e-6)


def test_config5_full_size_properties():
    """nu = 640 (V = 4 096 002, the size BASELINE config 5 names), fp32, fused assembly + step: a constant state is a
    fixed point of the freshly assembled operator on every deformed mesh (rows sum to zero by construction); two runs
    of the same five iterations agree bit for bit (no atomics anywhere in the assembly); the fused kernel and the
    two-kernel path agree; the edge length follows the deformation"""
    pts, tri = fa.make_icosphere(640, 0.1)
    mesh = fa.Mesh(pts, tri)
    part = mesh.partition(mesh.V // 256)
    om = part.ordered
    om.init_stats(fa.PREC_F64)
    dt = 0.08 * om.stats()["maxEdgeLen"] ** 2
    model = fa.default_model()
    plan = fa.Plan(om, fa.PREC_F32, part=part, ksteps=0, assemble=True)
    q = 2 * np.pi / 4.0
    m0 = fa.default_model()
    m0.r = 0.0
    n_const = np.full(om.V, 0.75)
    plan.set_state(n_const, n_const / (1 + n_const))
    for i in range(3):
        plan.deform(0.01, q, 0.3 * i)
        plan.assemble_step(dt, m0)
    n1, c1 = plan.get_state()
    assert np.array_equal(n1, n_const.astype(np.float32).astype(np.float64)) and np.abs(c1 - n_const / (1 + n_const)).max() <= 1e-6
    n0, c0 = fa.initial_condition(om.V, 2)
    runs = []
    for fused in (True, True, False):
        plan.set_state(n0, c0)
        edges = []
        for i in range(5):
            plan.deform(0.01, q, 0.3 * i)
            if fused:
                plan.assemble_step(dt, model)
            else:
                plan.assemble()
                plan.run(1, dt, model)
            edges.append(plan.max_edge())
        runs.append((plan.get_state(), edges))
    assert np.array_equal(runs[0][0][0], runs[1][0][0]) and np.array_equal(runs[0][0][1], runs[1][0][1])
    assert rel_l2(runs[0][0][0], runs[2][0][0]) <= 1e-6 and runs[0][1] == runs[2][1]
    assert len(set(runs[0][1])) == 5 and np.isfinite(runs[0][0][0]).all() and runs[0][0][0].min() >= 0
