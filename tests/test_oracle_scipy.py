"""Cross-checks the oracle against an INDEPENDENT numpy/scipy formulation of the same maths:
the P1 stiffness matrix assembled from element matrices (K_e = A * G^T G), the lumped mass from
element areas, the face gradient from the 3x3 least-norm solve per element, and the time step
written with sparse matrices.  No half-edges, no cotangent formulas, no corner conventions are
shared with oracle/fem_oracle.c, so agreement pins the restatement's arithmetic."""
import numpy as np
import pytest
import scipy.sparse as sp

import fem_amd as fa
import oracle_lib as ol
from helpers import rel_l2


def fem_matrices(pts, tri):
    V = len(pts)
    tri = tri.astype(np.int64)
    P = pts[tri]                                      # (F, 3, 3)
    e1, e2 = P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]
    nrm = np.cross(e1, e2)
    area = 0.5 * np.linalg.norm(nrm, axis=1)
    # hat-function gradients: solve [e1; e2; n] g = rhs per element (g in the face plane)
    A = np.stack([e1, e2, nrm], axis=1)               # (F, 3, 3)
    rhs = np.zeros((len(tri), 3, 3))                  # columns: phi_0, phi_1, phi_2
    rhs[:, 0, 0], rhs[:, 1, 0] = -1, -1
    rhs[:, 0, 1] = 1
    rhs[:, 1, 2] = 1
    G = np.linalg.solve(A, rhs)                       # (F, 3 xyz, 3 corners)
    Ke = area[:, None, None] * np.einsum("fxi,fxj->fij", G, G)
    rows = np.repeat(tri, 3, axis=1).ravel()
    cols = np.tile(tri, (1, 3)).ravel()
    K = sp.coo_matrix((Ke.ravel(), (rows, cols)), shape=(V, V)).tocsr()
    mass = np.bincount(tri.ravel(), np.repeat(area / 3, 3), V)
    L = sp.diags(-1.0 / mass) @ K                     # M^-1 * (-K): the reference's [wVtx | wHej]
    # interior angles from the law of cosines
    l = np.stack([np.linalg.norm(P[:, 2] - P[:, 1], axis=1), np.linalg.norm(P[:, 0] - P[:, 2], axis=1),
                  np.linalg.norm(P[:, 1] - P[:, 0], axis=1)], axis=1)     # opposite corner 0, 1, 2
    ang = np.arccos((np.roll(l, -1, 1) ** 2 + np.roll(l, -2, 1) ** 2 - l ** 2) / (2 * np.roll(l, -1, 1) * np.roll(l, -2, 1)))
    # vertex gradient operator: angle-weighted mean of face gradients
    theta = np.bincount(tri.ravel(), ang.ravel(), V)
    ops = []
    for x in range(3):
        vals = (ang[:, :, None] * G[:, x, None, :]).reshape(len(tri), 9)          # [corner v][corner j]
        r = np.repeat(tri, 3, axis=1).ravel()
        c = np.tile(tri, (1, 3)).ravel()
        ops.append(sp.diags(1.0 / theta) @ sp.coo_matrix((vals.ravel(), (r, c)), shape=(V, V)).tocsr())
    return L, ops, mass, area, ang


def scipy_step(L, ops, n, c, dt, model):
    nMax, D, r, alpha, S = model
    lapn, lapc = L @ n, L @ c
    dot = sum((op @ n) * (op @ c) for op in ops)
    dn = D * lapn - alpha * n * lapc - alpha * dot + S * r * n * (nMax - n)
    dc = lapc + S * (n / (1 + n) - c)
    return np.maximum(n + dt * dn, 0), np.maximum(c + dt * dc, 0)


@pytest.mark.parametrize("mesh", ["sphere", "strip"])
def test_operators_match_scipy(mesh):
    pts, tri = fa.make_icosphere(7, 0.1) if mesh == "sphere" else fa.make_strip(11, 41, 0.1)
    t = ol.Topology(len(pts), tri)
    s = ol.Stats(t, pts, ol.F64)
    L, ops, mass, area, ang = fem_matrices(pts, tri)
    # sparsity pattern: row i = {i} U 1-ring(i), ascending
    L.sort_indices()
    for v in range(t.V):
        cols = L.indices[L.indptr[v]:L.indptr[v + 1]]
        assert np.array_equal(cols[cols != v], t.col[t.rowptr[v]:t.rowptr[v + 1]])
    assert np.allclose(mass, s.areaPerVtx, rtol=1e-12) and np.allclose(area, s.meshArea, rtol=1e-12)
    # corner k of face f is the to-vertex of half-edge 3f + (k+2)%3
    assert np.allclose(ang, s.meshAngle[:3 * t.F].reshape(-1, 3)[:, [2, 0, 1]], atol=1e-12)
    rng = np.random.default_rng(0)
    f = rng.standard_normal(t.V)
    scale = np.abs(s.wVtx).max() * np.abs(f).max()
    assert np.abs(L @ f - s.laplacian(f)).max() <= 1e-12 * scale
    vg = s.vertex_gradients(s.face_gradients(f))
    for x in range(3):
        assert np.abs(ops[x] @ f - vg[:, x]).max() <= 1e-11 * np.abs(vg).max()


@pytest.mark.parametrize("mesh", ["sphere", "strip"])
def test_time_steps_match_scipy(mesh):
    pts, tri = fa.make_icosphere(7, 0.1) if mesh == "sphere" else fa.make_strip(11, 41, 0.1)
    t = ol.Topology(len(pts), tri)
    s = ol.Stats(t, pts, ol.F64)
    L, ops, *_ = fem_matrices(pts, tri)
    model = fa.default_model().as_array()
    dt = 0.1 * s.maxEdgeLen ** 2
    n0, c0 = ol.initial_condition(t.V, 5)
    n, c = n0.copy(), c0.copy()
    for _ in range(50):
        n, c = scipy_step(L, ops, n, c, dt, model)
    n_o, c_o = s.step(n0, c0, dt, 50, model)
    assert rel_l2(n_o, n) <= 1e-12 and rel_l2(c_o, c) <= 1e-12
    # the reference's own precision mix stays within fp32 round-off of it
    n_m, c_m = ol.Stats(t, pts, ol.MIXED).step(n0, c0, dt, 50, model)
    assert rel_l2(n_m, n) <= 1e-6 and rel_l2(c_m, c) <= 1e-6
    # alternative constant set of the reference's CPU path (main.cpp:49-54)
    alt = model.copy()
    alt[2], alt[3] = 1.52, 12.02
    n, c = n0.copy(), c0.copy()
    for _ in range(10):
        n, c = scipy_step(L, ops, n, c, dt, alt)
    n_o, c_o = s.step(n0, c0, dt, 10, alt)
    assert rel_l2(n_o, n) <= 1e-12 and rel_l2(c_o, c) <= 1e-12
