"""ctypes wrapper of oracle/libfem_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this; nothing under fem_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "libfem_oracle.so")

MIXED, F64 = "mixed", "f64"


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _load():
    src = [os.path.join(ORACLE_DIR, f) for f in ("fem_oracle.c", "fem_oracle_impl.inc")]
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in src if os.path.exists(s)):
        build()
    return C.CDLL(SO)


lib = _load()

u32p, i32p, i64p = C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
f64p, f32p = C.POINTER(C.c_double), C.POINTER(C.c_float)


def _p(a, t):
    return a.ctypes.data_as(t)


class Topology:
    """half-edge tables + CSR rows (fo_topology)"""

    def __init__(self, V, tri):
        tri = np.ascontiguousarray(tri, np.uint32).reshape(-1, 3)
        F = len(tri)
        n = max(6 * F, 1)
        opp, rowptr, row_he = np.empty(n, np.int32), np.empty(V + 1, np.int32), np.empty(n, np.int32)
        he_from, he_to, col = np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.uint32)
        H = C.c_int()
        rc = lib.fo_topology(V, F, _p(tri, u32p), _p(opp, i32p), _p(he_from, u32p), _p(he_to, u32p),
                             _p(rowptr, i32p), _p(col, u32p), _p(row_he, i32p), C.byref(H))
        if rc != 0:
            raise ValueError("fo_topology failed: %d" % rc)
        H = H.value
        self.V, self.F, self.H, self.tri = V, F, H, tri
        self.opp, self.he_from, self.he_to = opp[:H].copy(), he_from[:H].copy(), he_to[:H].copy()
        self.rowptr, self.col, self.row_he = rowptr, col[:H].copy(), row_he[:H].copy()


class Stats:
    """fo_assemble_{mixed,f64}: hot fields of MeshStats"""

    def __init__(self, topo, pts, prec=MIXED):
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        V, F, H = topo.V, topo.F, topo.H
        self.prec, self.topo, self.pts = prec, topo, pts
        gt, gp = (np.float32, f32p) if prec == MIXED else (np.float64, f64p)
        self.meshAngle, self.meshCot = np.empty(H), np.empty(H)
        self.meshArea, self.areaPerVtx = np.empty(F), np.empty(V)
        self.wHej, self.wVtx = np.empty(H), np.empty(V)
        self.gradVec12, self.gradVec13 = np.empty((F, 3), gt), np.empty((F, 3), gt)
        mx = C.c_double()
        fn = getattr(lib, "fo_assemble_" + prec)
        rc = fn(V, F, H, _p(pts, f64p), _p(topo.tri, u32p), _p(topo.opp, i32p), _p(topo.he_from, u32p),
                _p(topo.he_to, u32p), _p(self.meshAngle, f64p), _p(self.meshCot, f64p), _p(self.meshArea, f64p),
                _p(self.areaPerVtx, f64p), _p(self.wHej, f64p), _p(self.wVtx, f64p), C.byref(mx),
                _p(self.gradVec12, gp), _p(self.gradVec13, gp))
        assert rc == 0
        self.maxEdgeLen = mx.value
        self._gp = gp

    def laplacian(self, fn):
        t = self.topo
        fn = np.ascontiguousarray(fn, np.float64)
        out = np.empty(t.V)
        lib.fo_laplacian(t.V, _p(t.rowptr, i32p), _p(t.col, u32p), _p(t.row_he, i32p), _p(fn, f64p), _p(out, f64p),
                         _p(self.wHej, f64p), _p(self.wVtx, f64p))
        return out

    def face_gradients(self, fn):
        t = self.topo
        fn = np.ascontiguousarray(fn, np.float64)
        out = np.empty((t.F, 3), self.gradVec12.dtype)
        getattr(lib, "fo_face_gradients_" + self.prec)(t.F, _p(t.tri, u32p), _p(self.gradVec12, self._gp),
                                                        _p(self.gradVec13, self._gp), _p(fn, f64p), _p(out, self._gp))
        return out

    def vertex_gradients(self, face_grad):
        t = self.topo
        fg = np.ascontiguousarray(face_grad, self.gradVec12.dtype)
        out = np.empty((t.V, 3), self.gradVec12.dtype)
        getattr(lib, "fo_vertex_gradients_" + self.prec)(t.V, t.F, _p(t.rowptr, i32p), _p(t.row_he, i32p),
                                                          _p(fg, self._gp), _p(self.meshAngle, f64p), _p(out, self._gp))
        return out

    def step(self, n, c, dt, nsteps, model):
        """CPUrun restatement; returns new (n, c)"""
        t = self.topo
        n, c = np.array(n, np.float64), np.array(c, np.float64)
        model = np.ascontiguousarray(model, np.float64)
        rc = getattr(lib, "fo_step_" + self.prec)(
            t.V, t.F, _p(t.tri, u32p), _p(t.rowptr, i32p), _p(t.col, u32p), _p(t.row_he, i32p),
            _p(self.meshAngle, f64p), _p(self.wHej, f64p), _p(self.wVtx, f64p), _p(self.gradVec12, self._gp),
            _p(self.gradVec13, self._gp), _p(model, f64p), C.c_double(dt), C.c_int(nsteps), _p(n, f64p), _p(c, f64p))
        assert rc == 0
        return n, c


def set_threads(n):
    """threads of the OpenMP step loops (cpu_baseline leg); returns the count in effect"""
    return lib.fo_set_threads(C.c_int(n))


set_threads(1)          # tests run the oracle serially: results do not depend on the thread count,
                        # timing should not depend on the machine


def initial_condition(V, seed=1, nMax=1.0):
    n, c = np.empty(V), np.empty(V)
    lib.fo_initial_condition(C.c_int(V), C.c_uint32(seed), C.c_double(nMax), _p(n, f64p), _p(c, f64p))
    return n, c


def partition_apply(V, pts, tri, npart, n):
    """R1-R3 (partitioning_common.cpp:78-176) on a given npart[]"""
    tri = np.ascontiguousarray(tri, np.uint32).reshape(-1, 3)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    F = len(tri)
    perm_v = np.array(npart, np.int64)
    perm_f = np.empty(F, np.int64)
    lib.fo_face_parts(F, _p(tri, u32p), _p(perm_v, i64p), n, _p(perm_f, i64p))
    node_parts, element_parts = np.empty(n + 1, np.uint32), np.empty(n + 2, np.uint32)
    assert lib.fo_create_order_array(_p(node_parts, u32p), _p(perm_v, i64p), V, n) == 0
    assert lib.fo_create_order_array(_p(element_parts, u32p), _p(perm_f, i64p), F, n + 1) == 0
    pts_o, tri_o = np.empty_like(pts), np.empty_like(tri)
    lib.fo_reorder_mesh(V, F, _p(pts, f64p), _p(tri, u32p), _p(perm_v, i64p), _p(perm_f, i64p), _p(pts_o, f64p),
                        _p(tri_o, u32p))
    return perm_v, perm_f, node_parts, element_parts, pts_o, tri_o


def pack_ref_layout(stats, n, node_parts, element_parts):
    """halo faces + padded SoA arrays (main.cpp:207-342) from MIXED stats of the ordered mesh"""
    t = stats.topo
    assert stats.prec == MIXED
    node_parts = np.ascontiguousarray(node_parts, np.uint32)
    element_parts = np.ascontiguousarray(element_parts, np.uint32)
    halo_keys = np.empty(n + 1, np.uint32)
    halo_faces = np.empty(max(3 * t.F, 1), np.uint32)
    assert lib.fo_halo_faces(t.V, t.F, _p(t.tri, u32p), n, _p(node_parts, u32p), _p(element_parts, u32p),
                             _p(halo_keys, u32p), _p(halo_faces, u32p)) == 0
    maxV, W, maxF = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib.fo_pack_sizes(t.V, n, _p(node_parts, u32p), _p(element_parts, u32p), _p(halo_keys, u32p),
                      _p(t.rowptr, i32p), C.byref(maxV), C.byref(W), C.byref(maxF))
    maxV, W, maxF = maxV.value, W.value, maxF.value
    out = dict(halo_faces_keys=halo_keys, halo_faces=halo_faces[:halo_keys[n]].copy(),
               block_face_count=np.empty(n, np.uint32), vertex_weights=np.empty((n, maxV)),
               nbr_v=np.empty((n, W + 1, maxV), np.uint32), vv_weights=np.empty((n, W, maxV)),
               face_vertices=np.empty((n, 3, maxF), np.uint32), fv_weights_new=np.empty((n, 3, maxF)),
               he_grads=np.empty((n, 2, maxF, 4), np.float32), maxV=maxV, W=W, maxF=maxF)
    lib.fo_pack_ref_layout(t.V, t.F, _p(t.tri, u32p), n, _p(node_parts, u32p), _p(element_parts, u32p),
                           _p(halo_keys, u32p), _p(halo_faces, u32p), _p(t.rowptr, i32p), _p(t.col, u32p),
                           _p(t.row_he, i32p), _p(stats.meshAngle, f64p), _p(stats.wHej, f64p), _p(stats.wVtx, f64p),
                           _p(stats.gradVec12, f32p), _p(stats.gradVec13, f32p), maxV, W, maxF,
                           _p(out["block_face_count"], u32p), _p(out["vertex_weights"], f64p), _p(out["nbr_v"], u32p),
                           _p(out["vv_weights"], f64p), _p(out["face_vertices"], u32p),
                           _p(out["fv_weights_new"], f64p), _p(out["he_grads"], f32p))
    return out


def step_packed(lay, n_parts, node_parts, model, dt, fn):
    """literal CPU model of stepKernel on the packed layout (FEM_common.cu:82-166)"""
    node_parts = np.ascontiguousarray(node_parts, np.uint32)
    model = np.ascontiguousarray(model, np.float64)
    fn = np.ascontiguousarray(fn, np.float64)
    out = np.empty_like(fn)
    lib.fo_step_packed(n_parts, _p(node_parts, u32p), _p(lay["block_face_count"], u32p), lay["maxV"], lay["W"],
                       lay["maxF"], _p(lay["vertex_weights"], f64p), _p(lay["nbr_v"], u32p),
                       _p(lay["vv_weights"], f64p), _p(lay["face_vertices"], u32p), _p(lay["fv_weights_new"], f64p),
                       _p(lay["he_grads"], f32p), _p(model, f64p), C.c_double(dt), _p(fn, f64p), _p(out, f64p))
    return out


def colour_map(fn):
    fn = np.ascontiguousarray(fn, np.float64)
    out = np.empty((len(fn), 4))
    lib.fo_format(len(fn), _p(fn, f64p), _p(out, f64p))
    return out
