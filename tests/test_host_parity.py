"""Host side of the product (libfem_hip.so, no GPU needed) against the CPU oracle:
bit-exact for integer/index work (half-edge tables = CSR sparsity pattern, partition
renumbering, halo faces, packed index arrays), <= 1e-12 relative for the assembled weights."""
import os

import numpy as np
import pytest

import fem_amd as fa
import oracle_lib as ol
from helpers import Case

MESHES = {
    "ico1": lambda: fa.make_icosphere(1, 0.1),
    "ico9": lambda: fa.make_icosphere(9, 0.1),
    "strip": lambda: fa.make_strip(11, 41, 0.1),
    "strip2x2": lambda: fa.make_strip(2, 2, 0.1),
    "single": lambda: (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.0]]), np.array([[0, 1, 2]], np.uint32)),
}


@pytest.mark.parametrize("name", sorted(MESHES))
def test_topology_bit_exact(name):
    pts, tri = MESHES[name]()
    m = fa.Mesh(pts, tri)
    t = ol.Topology(m.V, tri)
    pt = m.topology()
    assert m.H == t.H
    for k in ("opp", "he_from", "he_to", "rowptr", "col", "row_he"):
        assert np.array_equal(pt[k], getattr(t, k)), k
    # CSR pattern properties: ascending unique columns, symmetric, no self loops
    for v in range(m.V):
        cols = pt["col"][pt["rowptr"][v]:pt["rowptr"][v + 1]]
        assert np.all(np.diff(cols.astype(np.int64)) > 0) and v not in cols
    assert np.array_equal(pt["he_from"][pt["opp"]], pt["he_to"])


@pytest.mark.parametrize("name", ["ico9", "strip", "single"])
@pytest.mark.parametrize("prec", [(ol.F64, fa.PREC_F64), (ol.MIXED, fa.PREC_MIXED)])
def test_mesh_stats_match_oracle(name, prec):
    pts, tri = MESHES[name]()
    m = fa.Mesh(pts, tri).init_stats(prec[1])
    so = ol.Stats(ol.Topology(m.V, tri), pts, prec[0])
    sp = m.stats()
    for k in ("meshAngle", "meshCot", "meshArea", "areaPerVtx", "wHej", "wVtx", "gradVec12", "gradVec13"):
        a, b = np.asarray(getattr(so, k), np.float64), sp[k]
        assert np.abs(a - b).max() <= 1e-12 * max(np.abs(a).max(), 1e-300), k
    assert sp["maxEdgeLen"] == so.maxEdgeLen
    # CPU operators kept by name (mesh_function.h:6-110)
    f = np.cos(3 * pts[:, 0]) + pts[:, 1] ** 2
    assert np.allclose(m.laplacian_cpu(f), so.laplacian(f), rtol=0, atol=1e-12 * np.abs(so.wVtx).max())
    fg = m.face_gradients_cpu(f)
    assert np.allclose(fg, so.face_gradients(f), rtol=0, atol=1e-12 * np.abs(fg).max())
    vg = m.vertex_gradients_cpu(fg)
    assert np.allclose(vg, so.vertex_gradients(so.face_gradients(f)), rtol=0, atol=1e-6 * np.abs(vg).max())


def test_mesh_errors():
    pts, tri = MESHES["ico1"]()
    bad = tri.copy()
    bad[0] = bad[0][::-1]                      # flipped face: a directed edge now occurs twice
    with pytest.raises(fa.FemError, match="manifold"):
        fa.Mesh(pts, bad)
    with pytest.raises(ValueError):
        ol.Topology(len(pts), bad)
    bad = tri.copy()
    bad[3, 1] = 99                             # index out of range
    with pytest.raises(fa.FemError):
        fa.Mesh(pts, bad)
    with pytest.raises(fa.FemError):
        fa.Mesh(pts, tri).stats()              # stats not initialised


@pytest.mark.parametrize("name,n", [("ico9", 7), ("strip", 3), ("ico1", 1), ("ico1", 12)])
def test_partition_contract_bit_exact(name, n):
    """R1-R3 of partitioning_common.cpp:78-176 on the product's own vertex parts"""
    pts, tri = MESHES[name]()
    m = fa.Mesh(pts, tri)
    npart = m.partition_vertices(n)
    assert npart.min() == 0 and npart.max() == n - 1
    sizes = np.bincount(npart, minlength=n)
    assert sizes.max() - sizes.min() <= 1                      # balanced split
    part = m.partition(n)
    pv, pf, nparts, eparts, pts_o, tri_o = ol.partition_apply(m.V, pts, tri, npart, n)
    assert np.array_equal(part.perm_v, pv) and np.array_equal(part.perm_f, pf)
    assert np.array_equal(part.node_parts, nparts) and np.array_equal(part.element_parts, eparts)
    assert np.array_equal(part.ordered.points, pts_o) and np.array_equal(part.ordered.tri, tri_o)
    # contract properties: permutations; vertices of part p are [node_parts[p], node_parts[p+1]);
    # faces in bucket p < n lie wholly inside part p, the last bucket holds the cut faces
    assert np.array_equal(np.sort(part.perm_v), np.arange(m.V))
    assert np.array_equal(np.sort(part.perm_f), np.arange(m.F))
    owner = np.searchsorted(part.node_parts, tri_o, side="right") - 1
    for p in range(n):
        fp = owner[part.element_parts[p]:part.element_parts[p + 1]]
        assert np.all(fp == p)
    cutf = owner[part.element_parts[n]:part.element_parts[n + 1]]
    assert np.all(cutf.min(axis=1) != cutf.max(axis=1)) if len(cutf) else True
    # an arbitrary (non-geometric) npart goes through the same contract
    rng = np.random.default_rng(1)
    rnd = rng.integers(0, n, m.V)
    rnd[:n] = np.arange(n)
    part2 = m.partition(n, npart=rnd)
    pv, pf, nparts, eparts, pts_o, tri_o = ol.partition_apply(m.V, pts, tri, rnd, n)
    assert np.array_equal(part2.perm_v, pv) and np.array_equal(part2.perm_f, pf)
    assert np.array_equal(part2.ordered.tri, tri_o) and np.array_equal(part2.element_parts, eparts)


def test_partition_golden_fixture():
    """regression pin of the RCB split (the stage METIS occupied): committed vertex parts of
    the nu=9 sphere (n=7) and the strip (n=3); generator: tests/golden/make_golden.py"""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "partition_golden.npz"))
    for name, n in (("ico9", 7), ("strip", 3)):
        pts, tri = MESHES[name]()
        assert np.array_equal(fa.Mesh(pts, tri).partition_vertices(n), g["%s_n%d" % (name, n)])


@pytest.mark.parametrize("name,n", [("ico9", 7), ("strip", 3), ("ico1", 1)])
def test_reference_layout_bit_exact(name, n):
    """initializeGPUData (main.cpp:207-342): halo faces and every padded array"""
    pts, tri = MESHES[name]()
    case = Case(pts, tri, n)
    case.mesh.init_stats(fa.PREC_MIXED)
    lay = fa.RefLayout(case.part)
    mine = lay.arrays()
    ref = ol.pack_ref_layout(case.ostats(ol.MIXED), n, case.part.node_parts, case.part.element_parts)
    assert (lay.maxV, lay.W, lay.maxF) == (ref["maxV"], ref["W"], ref["maxF"])
    for k in mine:
        if mine[k].size == 0:
            assert ref[k].size == 0, k
        elif mine[k].dtype.kind in "ui":
            assert np.array_equal(mine[k], ref[k]), k
        else:
            assert np.abs(mine[k] - ref[k]).max() <= 1e-12 * np.abs(ref[k]).max(), k
    # every face touching partition p appears exactly once among its interior+halo slots
    fv = mine["face_vertices"]
    for p in range(n):
        cnt = mine["block_face_count"][p]
        faces = {tuple(fv[p, :, s]) for s in range(cnt)}
        assert len(faces) == cnt
        lo, hi = case.part.node_parts[p], case.part.node_parts[p + 1]
        touching = {tuple(f) for f in case.tri if ((f >= lo) & (f < hi)).any()}
        assert faces == touching


def test_initial_condition_bit_exact():
    for seed in (0, 1, 12345):
        a, b = fa.initial_condition(5000, seed), ol.initial_condition(5000, seed)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_off_roundtrip(tmp_path):
    pts, tri = MESHES["ico9"]()
    rgb = (np.arange(3 * len(pts)) % 256).astype(np.uint8).reshape(-1, 3)
    path = str(tmp_path / "m.off")
    fa.off_write(path, pts, tri, rgb)
    p2, t2, c2 = fa.off_read(path)
    assert np.allclose(p2, pts, rtol=1e-8) and np.array_equal(t2, tri) and np.array_equal(c2, rgb)
    fa.off_write(path, pts, tri)
    p3, t3, c3 = fa.off_read(path)
    assert c3 is None and np.array_equal(t3, tri)
    # quads are fan-triangulated like a TriMesh import; comments and blank lines are skipped
    with open(path, "w") as f:
        f.write("# comment\nOFF\n4 1 0\n0 0 0\n1 0 0\n1 1 0\n\n0 1 0\n4 0 1 2 3\n")
    p4, t4, _ = fa.off_read(path)
    assert np.array_equal(t4, [[0, 1, 2], [0, 2, 3]])
    with pytest.raises(fa.FemError):
        fa.off_read(str(tmp_path / "missing.off"))
    with open(path, "w") as f:
        f.write("PLY\n")
    with pytest.raises(fa.FemError):
        fa.off_read(path)


def test_icosphere_generator():
    for nu in (1, 2, 3, 10):
        pts, tri = fa.make_icosphere(nu, 0.1)
        assert len(pts) == 10 * nu * nu + 2 and len(tri) == 20 * nu * nu
        deg = np.diff(fa.Mesh(pts, tri).topology()["rowptr"])
        assert (deg == 5).sum() == 12 and (deg == 6).sum() == len(pts) - 12
        # outward orientation: positive signed volume
        vol = np.einsum("ij,ij->i", pts[tri[:, 0]], np.cross(pts[tri[:, 1]], pts[tri[:, 2]])).sum() / 6
        assert vol > 0
        e = np.linalg.norm(pts[tri[:, 0]] - pts[tri[:, 1]], axis=1)
        assert 0.8 < e.mean() / 0.1 < 1.2


def test_scanned_like_generator():
    """BASELINE config 4 stand-in: irregular valences incl. local high-valence fans, still a closed
    oriented 2-manifold with usable triangles; seeded (same seed -> same mesh)."""
    pts, tri = fa.make_scanned_like(40, 0.1, seed=5)
    p2, t2 = fa.make_scanned_like(40, 0.1, seed=5)
    assert np.array_equal(pts, p2) and np.array_equal(tri, t2)
    p3, t3 = fa.make_scanned_like(40, 0.1, seed=6)
    assert not np.array_equal(tri, t3)
    V, F = len(pts), len(tri)
    assert V == 10 * 40 * 40 + 2 and F == 20 * 40 * 40
    m = fa.Mesh(pts, tri)                                   # raises if a directed edge repeats (non-manifold)
    topo = m.topology()
    assert m.H == 3 * F                                     # closed: no boundary half-edges
    deg = np.diff(topo["rowptr"])
    assert deg.min() >= 3 and deg.max() >= 10 and (deg > 6).mean() > 0.1 and (deg < 6).mean() > 0.1
    assert V - 3 * F // 2 + F == 2
    vol = np.einsum("ij,ij->i", pts[tri[:, 0]], np.cross(pts[tri[:, 1]], pts[tri[:, 2]])).sum() / 6
    r = np.linalg.norm(pts, axis=1)
    assert vol > 0.9 * 4 / 3 * np.pi * r.min() ** 3
    s = m.init_stats(fa.PREC_F64).stats()
    ang = np.degrees(s["meshAngle"][:3 * F])
    assert ang.min() > 5 and ang.max() < 165
    assert s["areaPerVtx"].min() > 0 and np.isfinite(s["wHej"]).all()
    # the oracle's topology agrees on the irregular mesh too
    t = ol.Topology(V, tri)
    for k in ("opp", "rowptr", "col", "row_he"):
        assert np.array_equal(topo[k], getattr(t, k)), k
    # flips/fans/noise off = the plain geodesic sphere
    p0, t0 = fa.make_scanned_like(7, 0.1, seed=1, flip_fraction=0, noise_amp=0, high_valence_fraction=0)
    pi, ti = fa.make_icosphere(7, 0.1)
    assert np.array_equal(t0, ti) and np.array_equal(p0, pi)
    with pytest.raises(fa.FemError):
        fa.make_scanned_like(7, 0.1, seed=1, noise_amp=0.7)


def test_off_reader_at_scale(tmp_path, monkeypatch):
    """the threaded line-split path of the OFF reader (>= 200 000 records): same arrays as the
    sequential parser, exact decimal conversion, comments / blank lines / CRLF / quads / colours,
    and the fallback for records that span lines"""
    pts, tri = fa.make_icosphere(100, 0.1)                       # 100 002 vertices + 200 000 faces
    rgb = (np.arange(3 * len(pts)) % 251).astype(np.uint8).reshape(-1, 3)
    path = str(tmp_path / "big.off")
    fa.off_write(path, pts, tri, rgb)
    text = open(path).read().split("\n")
    expect_pts = np.array([[float(x) for x in l.split()[:3]] for l in text[2:2 + len(pts)]])
    outs = []
    for threads in ("4", "1"):
        monkeypatch.setenv("FEM_HOST_THREADS", threads)
        p2, t2, c2 = fa.off_read(path)
        assert np.array_equal(p2, expect_pts) and np.array_equal(t2, tri) and np.array_equal(c2, rgb)
        outs.append(p2)
    # decorated variant: comments, blank lines, CRLF, float colours, one quad, face colours
    monkeypatch.setenv("FEM_HOST_THREADS", "4")
    V = len(pts)
    lines = ["COFF", "# made by test", "%d %d 0" % (V, len(tri) - 1), ""]
    for i in range(V):
        lines.append("%r %r %r %.6f %.6f %.6f 1.0\r" % (float(pts[i, 0]), float(pts[i, 1]), float(pts[i, 2]), rgb[i, 0] / 255,
                                                        rgb[i, 1] / 255, rgb[i, 2] / 255))
        if i % 30011 == 0:
            lines.append("   # a comment between records")
    a, b, c = (int(x) for x in tri[0])
    d = next(int(x) for x in tri[1] if int(x) not in (a, b, c))
    lines.append("4 %d %d %d %d  255 0 0" % (a, b, c, d))       # replaces the first two triangles
    for f in tri[2:]:
        lines.append("3 %d %d %d" % tuple(f))
    path2 = str(tmp_path / "decorated.off")
    open(path2, "w").write("\n".join(lines) + "\n")
    p3, t3, c3 = fa.off_read(path2)
    assert np.array_equal(p3, pts) and np.array_equal(c3, rgb)   # repr round-trips doubles exactly
    assert np.array_equal(t3[:2], [[a, b, c], [a, c, d]]) and np.array_equal(t3[2:], tri[2:])
    # records spanning lines: the sequential token parser takes over
    lines2 = text[:2] + [w for l in text[2:2 + V] for w in (" ".join(l.split()[:2]), " ".join(l.split()[2:]))] + text[2 + V:]
    path3 = str(tmp_path / "split.off")
    open(path3, "w").write("\n".join(lines2))
    p4, t4, c4 = fa.off_read(path3)
    assert np.array_equal(p4, expect_pts) and np.array_equal(t4, tri) and np.array_equal(c4, rgb)
    # a bad index in the middle of a big file is an error, not a crash
    bad = list(text)
    bad[2 + V + 150000] = "3 0 1 %d" % V
    open(path3, "w").write("\n".join(bad))
    with pytest.raises(fa.FemError):
        fa.off_read(path3)


def _components_per_part(mesh, npart, n):
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components
    topo = mesh.topology()
    rp, col = topo["rowptr"], topo["col"].astype(np.int64)
    row = np.repeat(np.arange(mesh.V), np.diff(rp))
    keep = npart[row] == npart[col]
    A = sp.csr_matrix((np.ones(keep.sum(), np.int8), (row[keep], col[keep])), shape=(mesh.V, mesh.V))
    ncomp, lab = connected_components(A, directed=False)
    return np.array([len(np.unique(lab[npart == p])) for p in range(n)])


def test_graph_partitioner():
    """coordinate-free graph bisection: balanced, deterministic, hierarchical ids,
    goes through the same R1-R3 contract; on a folded sheet -- two layers 1e-3 apart -- the
    coordinate split makes disconnected parts and a larger cut, the graph split does not"""
    pts, tri = fa.make_icosphere(20, 0.1)
    m = fa.Mesh(pts, tri)
    n = 13
    g1, g2 = m.partition_vertices(n, "graph"), m.partition_vertices(n, "graph")
    assert np.array_equal(g1, g2) and g1.min() == 0 and g1.max() == n - 1
    sizes = np.bincount(g1, minlength=n)
    assert sizes.max() - sizes.min() <= 1
    assert (_components_per_part(m, g1, n) == 1).all()
    part = m.partition(n, method="graph")
    pv, pf, nparts, eparts, pts_o, tri_o = ol.partition_apply(m.V, pts, tri, g1, n)
    assert np.array_equal(part.perm_v, pv) and np.array_equal(part.ordered.tri, tri_o)
    # quality in the range of the coordinate split on a geometric mesh
    assert part.cut <= 1.15 * m.partition(n).cut
    # hierarchical numbering: parts [0, n/2) and [n/2, n) are each one connected region
    half = (g1 >= n // 2).astype(np.int64)
    assert (_components_per_part(m, half, 2) == 1).all()
    # folded sheet
    nx, ny, h = 24, 160, 0.1
    sp_, st_ = fa.make_strip(nx, ny, h)
    iy = np.arange(nx * ny) % ny
    fold = sp_.copy()
    s = h / np.sqrt(2.0)
    yy = np.where(iy < ny // 2, iy, ny - 1 - iy) * s
    fold[:, 1], fold[:, 2] = yy, yy + np.where(iy < ny // 2, 0.0, 1e-3)
    fm = fa.Mesh(fold, st_)
    k = 16
    rcb, gr = fm.partition_vertices(k, "rcb"), fm.partition_vertices(k, "graph")
    assert (_components_per_part(fm, gr, k) == 1).all()
    assert (_components_per_part(fm, rcb, k) > 1).any()
    assert fm.partition(k, npart=gr).cut < 0.8 * fm.partition(k, npart=rcb).cut
    with pytest.raises(fa.FemError):
        m.partition_vertices(m.V + 1, "graph")


def test_subdivide_upsampler():
    """1-to-4 midpoint subdivision: counts, old vertices kept, area kept, orientation and
    boundary kept, still manifold; COFF pattern file upsampled two levels"""
    pts, tri = fa.make_strip(5, 7, 0.1)
    m0 = fa.Mesh(pts, tri).init_stats(fa.PREC_F64)
    E = m0.H // 2
    p1, t1 = fa.subdivide(pts, tri, 1)
    assert len(p1) == len(pts) + E and len(t1) == 4 * len(tri)
    assert np.array_equal(p1[:len(pts)], pts)
    m1 = fa.Mesh(p1, t1).init_stats(fa.PREC_F64)
    assert abs(m1.stats()["meshArea"].sum() - m0.stats()["meshArea"].sum()) <= 1e-12
    assert m1.H - 3 * m1.F == 2 * (m0.H - 3 * m0.F)                  # boundary edges doubled
    n0 = np.cross(pts[tri[:, 1]] - pts[tri[:, 0]], pts[tri[:, 2]] - pts[tri[:, 0]])
    n1 = np.cross(p1[t1[:, 1]] - p1[t1[:, 0]], p1[t1[:, 2]] - p1[t1[:, 0]])
    assert np.allclose(n1, np.repeat(n0, 4, axis=0) / 4)               # same orientation, quarter area
    ps, ts = fa.make_icosphere(3, 0.1)
    p2, t2 = fa.subdivide(ps, ts, 2)
    assert len(t2) == 16 * len(ts) and len(p2) - len(t2) // 2 == 2     # Euler characteristic of the sphere
    deg = np.diff(fa.Mesh(p2, t2).topology()["rowptr"])
    assert (deg == 5).sum() == 12 and (deg == 6).sum() == len(p2) - 12
    p0, t0 = fa.subdivide(ps, ts, 0)
    assert np.array_equal(p0, ps) and np.array_equal(t0, ts)
    bad = ts.copy()
    bad[0] = bad[0][::-1]
    with pytest.raises(fa.FemError):
        fa.subdivide(ps, bad, 1)


def test_ply_reader(tmp_path):
    """Stanford PLY, ascii and binary little-endian: mixed property types, colours, extra
    properties skipped, a quad fan-triangulated; malformed files are errors"""
    pts, tri = fa.make_icosphere(4, 0.1)
    V, F = len(pts), len(tri)
    rgb = (np.arange(3 * V) % 256).astype(np.uint8).reshape(-1, 3)
    # ascii: double coords, a skipped property, uchar colours, face with a leading scalar property
    a = str(tmp_path / "a.ply")
    with open(a, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by test\nelement vertex %d\nproperty double x\nproperty double y\n"
                "property double z\nproperty float confidence\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                "element face %d\nproperty uchar flags\nproperty list uchar int vertex_indices\nend_header\n" % (V, F - 1))
        for i in range(V):
            f.write("%r %r %r 0.5 %d %d %d\n" % (float(pts[i, 0]), float(pts[i, 1]), float(pts[i, 2]), *rgb[i]))
        q = [int(x) for x in tri[0]] + [next(int(x) for x in tri[1] if int(x) not in tri[0])]
        f.write("7 4 %d %d %d %d\n" % tuple(q))
        for t in tri[2:]:
            f.write("0 3 %d %d %d\n" % tuple(t))
    p2, t2, c2 = fa.ply_read(a)
    assert np.array_equal(p2, pts) and np.array_equal(c2, rgb)
    assert np.array_equal(t2[:2], [[q[0], q[1], q[2]], [q[0], q[2], q[3]]]) and np.array_equal(t2[2:], tri[2:])
    # binary little-endian: float coords, ushort extra property, uint8 list count, uint indices
    b = str(tmp_path / "b.ply")
    vrec = np.zeros(V, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("q", "<u2")])
    vrec["x"], vrec["y"], vrec["z"], vrec["q"] = pts[:, 0], pts[:, 1], pts[:, 2], 7
    frec = np.zeros(F, dtype=[("n", "u1"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4")])
    frec["n"], frec["a"], frec["b"], frec["c"] = 3, tri[:, 0], tri[:, 1], tri[:, 2]
    with open(b, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "property ushort q\nelement face %d\nproperty list uint8 uint vertex_index\nend_header\n" % (V, F)).encode())
        f.write(vrec.tobytes())
        f.write(frec.tobytes())
    p3, t3, c3 = fa.read_mesh(b)
    assert c3 is None and np.array_equal(t3, tri) and np.array_equal(p3, pts.astype(np.float32).astype(np.float64))
    fa.Mesh(p3, t3)                                            # a valid manifold mesh
    # truncated body, big-endian, bad index, not a ply
    open(str(tmp_path / "c.ply"), "wb").write(open(b, "rb").read()[:-5])
    open(str(tmp_path / "d.ply"), "wb").write(open(b, "rb").read().replace(b"binary_little_endian", b"binary_big_endian"))
    open(str(tmp_path / "e.ply"), "w").write(open(a).read().replace("\n0 3 ", "\n0 3 99999 ", 1))
    open(str(tmp_path / "f.ply"), "w").write("OFF\n0 0 0\n")
    for name in "cdef":
        with pytest.raises(fa.FemError):
            fa.ply_read(str(tmp_path / (name + ".ply")))


def test_shard_restricted_assembly_rows_are_bit_identical():
    """fem_mesh_init_stats_shard: the rows of a shard's local vertices equal the full assembly's bit for bit, nothing
    else is assembled, maxEdgeLen stays the whole mesh's (dt must agree on every rank)"""
    pts, tri = fa.make_scanned_like(24, 0.1, seed=3, high_valence_fraction=0.01)
    part = fa.Mesh(pts, tri).partition(30)
    om = part.ordered
    for prec in (fa.PREC_F64, fa.PREC_MIXED):
        om.init_stats(prec)
        full = om.stats()
        shard = fa.Shard(part, 1, 3, 4)
        om.init_stats_shard(prec, shard)
        loc = om.stats()
        l2g = shard.loc2glob.astype(np.int64)
        assert len(l2g) < om.V
        for key in ("wVtx", "areaPerVtx"):
            assert np.array_equal(full[key][l2g], loc[key][l2g])
            rest = np.ones(om.V, bool)
            rest[l2g] = False
            assert (loc["wVtx"][rest] == 0).all()
        rows = np.isin(om.topology()["he_from"], l2g)
        assert np.array_equal(full["wHej"][rows], loc["wHej"][rows]) and (loc["wHej"][~rows] == 0).all()
        assert loc["maxEdgeLen"] == full["maxEdgeLen"]


def test_readers_refuse_counts_the_file_cannot_hold(tmp_path):
    """a header that promises billions of records in a few bytes is an IO error, not an allocation of that size
    (fem_hip.h: no entry point aborts)"""
    off = tmp_path / "liar.off"
    off.write_text("OFF\n4000000000 4000000000 0\n0 0 0\n")
    with pytest.raises(fa.FemError):
        fa.off_read(str(off))
    ply = tmp_path / "liar.ply"
    ply.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                    b"element face 2000000000\nproperty list uchar int vertex_indices\nend_header\n" + b"\0" * 36)
    with pytest.raises(fa.FemError):
        fa.ply_read(str(ply))
    ply.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 3000000000\nproperty float x\nproperty float y\nproperty float z\n"
                    b"element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n")
    with pytest.raises(fa.FemError):
        fa.ply_read(str(ply))


def test_multilevel_dual_graph_partition():
    """the METIS_PartMeshDual stage as a multilevel k-way partition of the dual graph: every part non-empty, sizes
    within a few per cent of the mean (the per-block caps as a balance constraint), deterministic, the element
    partition it was induced from is balanced and its dual cut is what dualEdgeCut counts, the cut is within 15 % of
    the coordinate bisection's, and the deterministic stage after it (R1-R3) accepts the result"""
    import ctypes as C
    for (pts, tri), n in ((fa.make_icosphere(48, 0.1), 24), (fa.make_scanned_like(40, 0.1, seed=3, high_valence_fraction=0.01), 20),
                          (fa.make_strip(11, 41, 0.1), 3)):
        mesh = fa.Mesh(pts, tri)
        npart = mesh.partition_vertices(n, "multilevel")
        again = mesh.partition_vertices(n, "multilevel")
        assert np.array_equal(npart, again)
        sizes = np.bincount(npart, minlength=n)
        assert sizes.min() > 0 and npart.min() == 0 and npart.max() == n - 1
        assert sizes.max() <= 1.06 * mesh.V / n + 2 and sizes.min() >= 0.90 * mesh.V / n - 2
        epart = np.empty(mesh.F, np.int64)
        cut = C.c_int64()
        fa._check(fa.lib.fem_partition_vertices_multilevel(mesh.h, n, npart.ctypes.data_as(fa.i64p), epart.ctypes.data_as(fa.i64p),
                                                            C.byref(cut)))
        fsz = np.bincount(epart, minlength=n)
        assert fsz.min() > 0 and fsz.max() <= 1.04 * mesh.F / n + 2
        # dual edges (elements sharing an edge: ncommon = 2) between different parts, counted independently
        t = mesh.topology()
        F3 = 3 * mesh.F
        h = np.arange(F3)
        ho = t["opp"][:F3]
        inner = (ho > h) & (ho < F3)
        assert int((epart[h[inner] // 3] != epart[ho[inner] // 3]).sum()) == cut.value
        part_ml = mesh.partition(n, npart=npart)
        part_rcb = mesh.partition(n)
        assert part_ml.cut <= 1.15 * part_rcb.cut
        assert part_ml.node_parts[-1] == mesh.V and np.array_equal(np.diff(part_ml.node_parts.astype(np.int64)), sizes)
    with pytest.raises(fa.FemError):
        fa.Mesh(*fa.make_icosphere(2, 0.1)).partition_vertices(100000, "multilevel")


def rolled_sheet(nx=600, ny=120, h=0.1, gap=0.35):
    """a sheet rolled into a spiral whose layers lie closer in space (gap) than a part is wide: near in coordinates,
    far on the surface -- the kind of surface a coordinate split cuts into stacks of disconnected pieces"""
    u = np.arange(nx) * h
    b = gap / (2 * np.pi)
    theta = np.sqrt(2 * (u + 5.0) / b)
    r = b * theta
    x, z = r * np.cos(theta), r * np.sin(theta)
    pts = np.stack([np.repeat(x[:, None], ny, 1).ravel(), np.repeat((np.arange(ny) * h)[None, :], nx, 0).ravel(),
                    np.repeat(z[:, None], ny, 1).ravel()], 1)
    idx = np.arange(nx * ny).reshape(nx, ny)
    a, b_, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    return pts, np.concatenate([np.stack([a, b_, c], 1), np.stack([a, c, d], 1)]).astype(np.uint32)


def test_auto_partitioner_keeps_rcb_unless_coordinates_mislead():
    """stage R0 as the drivers run it (FEM_PARTITIONER=auto): identical to the coordinate bisection on the sphere and on
    the irregular scanned-like mesh; on a rolled sheet it switches to the coordinate-free graph bisection, whose parts
    are connected (half the edge cut), with the same exact balance -- and the R1-R3 contract holds on its output"""
    for pts, tri in (fa.make_icosphere(48, 0.1), fa.make_scanned_like(32, 0.1, seed=3)):
        m = fa.Mesh(pts, tri)
        n = m.V // 400
        assert np.array_equal(m.partition_vertices(n, "auto"), m.partition_vertices(n, "rcb")) and m.last_partitioner == "rcb"
    pts, tri = rolled_sheet()
    m = fa.Mesh(pts, tri)
    n = m.V // 600
    auto = m.partition_vertices(n, "auto")
    assert m.last_partitioner == "graph" and np.array_equal(auto, m.partition_vertices(n, "graph"))
    sizes = np.bincount(auto, minlength=n)
    assert sizes.max() - sizes.min() <= 1
    cut_auto, cut_rcb = m.partition(n, npart=auto).cut, m.partition(n, method="rcb").cut
    assert cut_auto < 0.6 * cut_rcb, (cut_auto, cut_rcb)
    part = m.partition(n, method="auto")
    pv, pf, nparts, eparts, pts_o, tri_o = ol.partition_apply(m.V, pts, tri, auto, n)
    assert np.array_equal(part.perm_v, pv) and np.array_equal(part.perm_f, pf) and np.array_equal(part.node_parts, nparts)
    assert np.array_equal(part.element_parts, eparts) and np.array_equal(part.ordered.tri, tri_o)


def test_wide_rows_first_numbering():
    """fem_mesh_wide_rows_first (pre-pass of the partition stage for irregular meshes): a stable renumbering by the
    number of row entries beyond 6, most first -- the same surface (every vertex keeps its point, faces keep their order
    and corner order), the assembled operator rows merely relabelled; a mesh without wide rows is left alone; and after
    the partition stage (stable inside a part, partitioning_common.cpp:112-139) the wide rows lead every part, widest first"""
    pts, tri = fa.make_scanned_like(32, 0.1, seed=3)
    m = fa.Mesh(pts, tri)
    rp = m.topology()["rowptr"].astype(np.int64)
    extra = np.maximum(np.diff(rp) - 6, 0)
    m2, new = m.wide_rows_first(6)
    order = np.argsort(-extra, kind="stable")
    expect = np.empty(m.V, np.int64)
    expect[order] = np.arange(m.V)
    assert np.array_equal(new, expect)
    assert np.array_equal(m2.points[new], pts) and np.array_equal(m2.tri, new[tri.astype(np.int64)].astype(np.uint32))
    m.init_stats(fa.PREC_F64)
    m2.init_stats(fa.PREC_F64)
    s1, s2 = m.stats(), m2.stats()
    assert np.array_equal(s2["wVtx"][new], s1["wVtx"]) and np.array_equal(s2["areaPerVtx"][new], s1["areaPerVtx"])
    assert np.array_equal(s2["meshArea"], s1["meshArea"]) and s1["maxEdgeLen"] == s2["maxEdgeLen"]
    same, none = fa.Mesh(*fa.make_icosphere(8, 0.1)).wide_rows_first(6)
    assert none is None
    # through the drivers' partition stage: inside every part the extras counts never increase
    from fem_amd import dist as fdist
    part = fdist._partition(m, 20)
    om = part.ordered
    assert np.array_equal(om.points[part.perm_v], pts)          # perm_v refers to the mesh the caller passed
    ex = np.maximum(np.diff(om.topology()["rowptr"].astype(np.int64)) - 6, 0)
    assert ex.max() > 0
    for p in range(part.n):
        seg = ex[part.node_parts[p]:part.node_parts[p + 1]]
        assert np.all(np.diff(seg) <= 0)


def test_spectrum_bound_against_dense_eigenvalues():
    """fem_mesh_spectrum_bound (the interval of the Chebyshev solver of the semi-implicit step): never below the largest
    eigenvalue of -L -- dense eigenvalues of the assembled rows on a sphere, an open strip, a scan-like mesh with
    valences 3..16 and one with obtuse triangles and negative cotangent weights, where the old bound (twice the largest
    row SUM) would not have been one -- never above Gershgorin's; on the geodesic sphere (triangle sizes vary by the
    usual class-I factor) 1.19 x the true value where Gershgorin is 1.59 x"""
    import fem_amd as fa
    from helpers import sphere_case, strip_case, scan_case, pathological_scan
    from helpers import stretched_sheet
    sq_pts, sq_tri = stretched_sheet(9)
    ps_pts, ps_tri = pathological_scan(10)
    meshes = [("sphere", fa.Mesh(*fa.make_icosphere(8, 0.1))), ("strip", strip_case(3).mesh), ("scan", scan_case(10, 4).mesh),
              ("stretched", fa.Mesh(sq_pts, sq_tri)), ("pathological", fa.Mesh(ps_pts, ps_tri))]
    for name, mesh in meshes:
        mesh.init_stats(fa.PREC_F64)
        V = mesh.V
        L = np.empty((V, V))
        for v in range(V):
            e = np.zeros(V); e[v] = 1.0
            L[:, v] = mesh.laplacian_cpu(e)
        L = L[np.isfinite(L).all(axis=1)][:, np.isfinite(L).all(axis=0)] if not np.isfinite(L).all() else L
        lam = float(np.max(-np.linalg.eigvals(L).real))
        gersh = float(np.max(np.abs(np.diag(L)) + (np.abs(L).sum(axis=1) - np.abs(np.diag(L)))))
        b = mesh.spectrum_bound()
        assert lam <= b * (1 + 1e-12) and b <= gersh * (1 + 1e-12), (name, lam, b, gersh)
        if name == "sphere":
            assert b <= 1.25 * lam and gersh >= 1.5 * lam, (lam, b, gersh)
        if name == "stretched":
            assert (L - np.diag(np.diag(L))).min() < 0           # negative weights are present
            assert 2 * np.max(np.abs(np.diag(L))) < gersh        # ... and twice the largest row sum is not Gershgorin's bound
