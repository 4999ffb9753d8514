"""Pins the CPU oracle by closed-form known answers (the reference has no
golden vectors, so these are what the restatement is anchored on)."""
import numpy as np
import pytest

import fem_amd as fa
import oracle_lib as ol

PRECS = [ol.F64, ol.MIXED]


def tol(prec, t64=1e-12, t32=2e-5):
    return t64 if prec == ol.F64 else t32


@pytest.mark.parametrize("prec", PRECS)
def test_equilateral_triangle(prec):
    h = 0.37
    pts = np.array([[0, 0, 0], [h, 0, 0], [h / 2, h * np.sqrt(3) / 2, 0]])
    tri = np.array([[0, 1, 2]], np.uint32)
    t = ol.Topology(3, tri)
    assert t.H == 6 and (t.opp[:3] >= 3).all()          # 3 interior + 3 boundary half-edges
    s = ol.Stats(t, pts, prec)
    e = tol(prec)
    assert np.allclose(s.meshAngle[:3], np.pi / 3, atol=e) and np.all(s.meshAngle[3:] == 0)
    assert np.allclose(s.meshCot[:3], 1 / np.sqrt(3), atol=10 * e) and np.all(s.meshCot[3:] == 0)
    assert np.isclose(s.meshArea[0], np.sqrt(3) / 4 * h * h, rtol=e)
    assert np.allclose(s.areaPerVtx, s.meshArea[0] / 3, rtol=e)
    assert np.isclose(s.maxEdgeLen, h, rtol=e)
    # every directed edge sees one cot(60 deg): w = (1/sqrt3) / (2 A/3)
    assert np.allclose(s.wHej, (1 / np.sqrt(3)) / (2 * s.meshArea[0] / 3), rtol=10 * e)
    assert np.allclose(s.wVtx, -2 * s.wHej[0], rtol=10 * e)
    # hat-function gradients: |grad| = 1/height, gradVec12 = grad phi_a, gradVec13 = grad phi_b
    height = h * np.sqrt(3) / 2
    assert np.allclose(np.linalg.norm(s.gradVec12, axis=1), 1 / height, rtol=10 * e)
    g_a = s.gradVec12[0].astype(np.float64)
    assert np.isclose(g_a @ (pts[0] - pts[1]), 1.0, rtol=10 * e)   # phi_a goes 0 -> 1 from b to a
    assert abs(g_a @ (pts[2] - pts[1])) < 10 * e                    # and is constant along b-c


@pytest.mark.parametrize("prec", PRECS)
def test_strip_is_five_point_stencil(prec):
    """right-isoceles grid with the connectivity of ResultantPattern.off:454-457: the diagonal
    edges face 90-degree angles (cot = 0), so interior rows are the 5-point stencil / h^2"""
    h = 0.1
    pts, tri = fa.make_strip(11, 41, h)
    t = ol.Topology(len(pts), tri)
    s = ol.Stats(t, pts, prec)
    e = tol(prec, 1e-9, 2e-4)
    ny = 41
    deg = np.diff(t.rowptr)
    for ix in (1, 5, 9):
        for iy in (1, 20, 39):
            v = ix * ny + iy
            assert deg[v] == 6
            row = {int(t.col[k]): s.wHej[t.row_he[k]] for k in range(t.rowptr[v], t.rowptr[v + 1])}
            for nb in (v - 1, v + 1, v - ny, v + ny):
                assert np.isclose(row[nb], 1 / h ** 2, rtol=e)
            for nb in (v + ny + 1, v - ny - 1):
                assert abs(row[nb]) < e / h ** 2
            assert np.isclose(s.wVtx[v], -4 / h ** 2, rtol=e)
            assert np.isclose(s.areaPerVtx[v], h * h, rtol=e)
    assert np.isclose(s.meshArea.sum(), 1.0 * 4.0, rtol=e)          # 10h x 40h sheet


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("mesh", ["strip", "sphere"])
def test_constant_and_linear_fields(prec, mesh):
    if mesh == "strip":
        pts, tri = fa.make_strip(7, 9, 0.1)
    else:
        pts, tri = fa.make_icosphere(6, 0.1)
    t = ol.Topology(len(pts), tri)
    s = ol.Stats(t, pts, prec)
    e = tol(prec, 1e-10, 5e-4)
    scale = np.abs(s.wVtx).max()
    # rows of [wVtx | wHej] sum to zero; lumped masses sum to the area
    rowsum = s.wVtx + np.add.reduceat(s.wHej[t.row_he], t.rowptr[:-1])
    assert np.abs(rowsum).max() <= 1e-12 * scale
    assert np.isclose(s.areaPerVtx.sum(), s.meshArea.sum(), rtol=1e-12)
    const = np.full(t.V, 3.25)
    assert np.abs(s.laplacian(const)).max() <= 1e-11 * scale
    assert np.abs(s.face_gradients(const)).max() == 0
    assert np.abs(s.vertex_gradients(s.face_gradients(const))).max() == 0
    if mesh == "strip":
        # linear field on a planar mesh: face and vertex gradients exact, laplacian 0 inside
        d = np.array([0.3, -1.1, 0.7])
        normal = np.array([0.0, -1.0, 1.0]) / np.sqrt(2)
        d_tan = d - (d @ normal) * normal
        f = pts @ d
        fg = s.face_gradients(f).astype(np.float64)
        assert np.abs(fg - d_tan).max() <= e * 10
        vg = s.vertex_gradients(fg.astype(s.gradVec12.dtype)).astype(np.float64)
        assert np.abs(vg - d_tan).max() <= e * 10
        lap = s.laplacian(f)
        interior = np.diff(t.rowptr) == 6
        assert np.abs(lap[interior]).max() <= max(1e-9, e) * scale * np.abs(f).max()


@pytest.mark.parametrize("prec", PRECS)
def test_sphere_area_and_harmonics(prec):
    """sum of areas -> 4 pi R^2; coordinate functions are eigenfunctions of the sphere's
    Laplace-Beltrami operator with eigenvalue -2/R^2 (second-order accurate check of A1-A3)"""
    pts, tri = fa.make_icosphere(24, 0.1)
    R = np.linalg.norm(pts[0])
    assert np.allclose(np.linalg.norm(pts, axis=1), R, rtol=1e-13)
    t = ol.Topology(len(pts), tri)
    assert t.H == 3 * t.F                                 # closed: no boundary half-edges
    assert t.V - t.H // 2 + t.F == 2                      # Euler characteristic of the sphere
    s = ol.Stats(t, pts, prec)
    assert 0.995 < s.meshArea.sum() / (4 * np.pi * R * R) < 1.0
    z = pts[:, 2].copy()
    err = s.laplacian(z) + 2 / R ** 2 * z
    # barycentric lumping is only weakly consistent: O(1) pointwise error at the 12 valence-5
    # vertices, first-order convergence in the area-weighted L2 norm, exact in the weak form
    A = s.areaPerVtx
    assert np.sqrt((A * err ** 2).sum() / (A * (2 / R ** 2 * z) ** 2).sum()) <= 0.006
    assert abs((A * err * z).sum()) <= tol(prec, 1e-9, 1e-4) * (A * z * z).sum()
    regular = np.diff(t.rowptr) == 6
    assert np.abs(err[regular]).max() <= 0.03 * 2 / R
    # tangential gradient of z: e_z - (e_z . n) n
    vg = s.vertex_gradients(s.face_gradients(pts[:, 2].copy())).astype(np.float64)
    nrm = pts / R
    exact = np.array([0, 0, 1.0]) - nrm[:, 2:3] * nrm
    assert np.abs(vg - exact).max() <= 0.02


def test_initial_condition_law():
    """main.cpp:182-186: beta in [-betaMax/2, betaMax/2), n = nMax - beta, c = 1/(1+n)"""
    n, c = ol.initial_condition(10000, seed=3)
    assert np.all(np.abs(1 - n) <= 0.005 + 1e-9) and np.array_equal(c, 1 / (1 + n))
    assert abs(n.mean() - 1) < 2e-4 and np.std(n) > 0.002
    n2, _ = ol.initial_condition(10000, seed=3)
    n3, _ = ol.initial_condition(10000, seed=4)
    assert np.array_equal(n, n2) and not np.array_equal(n, n3)


def test_step_fixed_points_and_clamp():
    pts, tri = fa.make_icosphere(4, 0.1)
    t = ol.Topology(len(pts), tri)
    model = fa.default_model().as_array()
    for prec in PRECS:
        s = ol.Stats(t, pts, prec)
        dt = 0.1 * s.maxEdgeLen ** 2
        # homogeneous steady state (n, c) = (nMax, nMax/(1+nMax)) is a fixed point
        n, c = s.step(np.ones(t.V), np.full(t.V, 0.5), dt, 5, model)
        assert np.abs(n - 1).max() < 1e-12 and np.abs(c - 0.5).max() < 1e-12
        # states driven negative are clamped to exactly 0 (main.cpp:660-664 / FEM_common.cu:164-165)
        n0 = np.ones(t.V)
        n0[::3] = 0.0
        n, c = s.step(n0, np.full(t.V, 0.5), 100 * dt, 1, model)
        assert (n == 0).any() and (n >= 0).all() and (c >= 0).all()


def test_colour_map():
    """formatKernel, FEM_common.cu:183-194"""
    fn = np.array([[2.0, 9], [3.0, 9], [2.5, 9]])
    rgba = ol.colour_map(fn)
    assert np.allclose(rgba, [[0, 1, 1, 1], [1, 0, 0, 1], [0.5, 0.5, 0.5, 1]])
