"""fem_amd -- ctypes plumbing over libfem_hip.so (the C-ABI in include/fem_hip.h).

The product is the shared library (host C++ + hand-written gfx950 kernels);
this module only binds it for tests, bench.py and Python callers.  There is
no CPU fallback: importing fails loudly when the library has not been built
(`make -C fem_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfem_hip.so")

PREC_MIXED, PREC_F64, PREC_F32 = 0, 1, 2

# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same soname as
# /opt/rocm's).  If torch is imported AFTER this library has pulled in /opt/rocm's copy, a second
# runtime gets loaded and torch sees no GPU.  Importing torch first makes the dynamic loader
# reuse torch's copy for libfem_hip.so as well.  Without torch the system runtime is used.
try:
    import torch  # noqa: F401
except ImportError:
    pass

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "fem_amd: %s is missing -- build it first (make -C fem_amd/csrc). "
        "There is no fallback path." % LIB_PATH)
lib = C.CDLL(LIB_PATH)

u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p


class Model(C.Structure):
    """fem_model: constants of reference src/FEM_common.cu:9-13 as data."""
    _fields_ = [("nMax", C.c_double), ("D", C.c_double), ("r", C.c_double),
                ("alpha", C.c_double), ("S", C.c_double)]

    def as_array(self):
        return np.array([self.nMax, self.D, self.r, self.alpha, self.S])


class FemError(RuntimeError):
    pass


lib.fem_error_string.restype = C.c_char_p
lib.fem_kernel_build.restype = C.c_char_p
lib.fem_error_detail.restype = C.c_char_p
lib.fem_plan_operator_bytes.restype = C.c_uint64
lib.fem_plan_operator_bytes.argtypes = [vp]
lib.fem_plan_operator_format.restype = C.c_int
lib.fem_plan_operator_format.argtypes = [vp]
lib.fem_free.argtypes = [vp]
lib.fem_plan_pack_halo.argtypes = [vp, C.c_uint64, vp]
lib.fem_plan_pack_halo_from.argtypes = [vp, C.c_int, C.c_uint64, vp]
lib.fem_plan_launch_tiles.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.c_double, vp, vp]
lib.fem_plan_format_noalloc.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint64, vp]
lib.fem_plan_coef_count.restype = C.c_uint64
lib.fem_plan_coef_count.argtypes = [vp]
for _name in ("fem_mesh_destroy", "fem_ref_layout_destroy", "fem_ref_device_destroy", "fem_plan_destroy",
              "fem_shard_destroy"):
    getattr(lib, _name).argtypes = [vp]
    getattr(lib, _name).restype = None


def _check(rc):
    if rc < 0:
        raise FemError("%s (%d): %s" % (lib.fem_error_string(rc).decode(), rc,
                                        lib.fem_error_detail().decode()))
    return rc


def _ptr(a, t):
    return a.ctypes.data_as(t)


def kernel_build():
    """digest of the device sources the loaded library was built from (fem_kernel_build)"""
    return lib.fem_kernel_build().decode()


def default_model():
    m = Model()
    lib.fem_default_model(C.byref(m))
    return m


def _take(ptr, n, dtype):
    """copy a malloc'ed C array into numpy and free it"""
    out = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    lib.fem_free(C.cast(ptr, vp))
    return out


def _mesh_out(fn, *args):
    V, F = C.c_int(), C.c_int()
    pts, tri = f64p(), u32p()
    _check(fn(*args, C.byref(V), C.byref(F), C.byref(pts), C.byref(tri)))
    return (_take(pts, 3 * V.value, np.float64).reshape(-1, 3),
            _take(tri, 3 * F.value, np.uint32).reshape(-1, 3))


def make_icosphere(nu, edge=0.1):
    return _mesh_out(lib.fem_make_icosphere, C.c_int(nu), C.c_double(edge))


def make_scanned_like(nu, edge=0.1, seed=1, flip_fraction=0.1, noise_amp=0.1, high_valence_fraction=0.001,
                      high_valence=16):
    return _mesh_out(lib.fem_make_scanned_like, C.c_int(nu), C.c_double(edge), C.c_uint32(seed),
                     C.c_double(flip_fraction), C.c_double(noise_amp), C.c_double(high_valence_fraction),
                     C.c_int(high_valence))


def subdivide(points, tri, levels=1):
    """1-to-4 midpoint subdivision, `levels` rounds (the upsampler for real scans)"""
    points, tri = np.ascontiguousarray(points, np.float64), np.ascontiguousarray(tri, np.uint32)
    return _mesh_out(lib.fem_subdivide, C.c_int(len(points)), C.c_int(len(tri)), _ptr(points, f64p), _ptr(tri, u32p),
                     C.c_int(levels))


def make_strip(nx=11, ny=41, h=0.1):
    return _mesh_out(lib.fem_make_strip, C.c_int(nx), C.c_int(ny), C.c_double(h))


def ply_read(path):
    return off_read(path, _fn=lib.fem_ply_read)


def read_mesh(path):
    """.off / .ply by extension"""
    return ply_read(path) if path.lower().endswith(".ply") else off_read(path)


def off_read(path, _fn=None):
    V, F = C.c_int(), C.c_int()
    pts, tri, rgb = f64p(), u32p(), u8p()
    _check((_fn or lib.fem_off_read)(path.encode(), C.byref(V), C.byref(F), C.byref(pts), C.byref(tri), C.byref(rgb)))
    colors = _take(rgb, 3 * V.value, np.uint8).reshape(-1, 3) if rgb else None
    return (_take(pts, 3 * V.value, np.float64).reshape(-1, 3),
            _take(tri, 3 * F.value, np.uint32).reshape(-1, 3), colors)


def off_write(path, points, tri, rgb=None):
    points = np.ascontiguousarray(points, np.float64)
    tri = np.ascontiguousarray(tri, np.uint32)
    rgbp = None
    if rgb is not None:
        rgb = np.ascontiguousarray(rgb, np.uint8)
        rgbp = _ptr(rgb, u8p)
    _check(lib.fem_off_write(path.encode(), len(points), len(tri), _ptr(points, f64p), _ptr(tri, u32p), rgbp))


def initial_condition(V, seed=1, nMax=1.0):
    n, c = np.empty(V), np.empty(V)
    lib.fem_initial_condition(C.c_int(V), C.c_uint32(seed), C.c_double(nMax), _ptr(n, f64p), _ptr(c, f64p))
    return n, c


class Mesh:
    """fem_mesh handle: points + faces + half-edge tables (+ stats once initialised)."""

    def __init__(self, points=None, tri=None, handle=None):
        if handle is None:
            points = np.ascontiguousarray(points, np.float64)
            tri = np.ascontiguousarray(tri, np.uint32)
            h = vp()
            _check(lib.fem_mesh_create(len(points), len(tri), _ptr(points, f64p), _ptr(tri, u32p), C.byref(h)))
            handle = h
        self.h = handle
        V, F, H = C.c_int(), C.c_int(), C.c_int()
        lib.fem_mesh_counts(self.h, C.byref(V), C.byref(F), C.byref(H))
        self.V, self.F, self.H = V.value, F.value, H.value
        self.precision = None

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.fem_mesh_destroy(self.h)
            self.h = None

    def _view(self, ptr, n, shape=None):
        a = np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0)
        return a.reshape(shape) if shape else a

    @property
    def points(self):
        p, t = f64p(), u32p()
        lib.fem_mesh_arrays(self.h, C.byref(p), C.byref(t))
        return self._view(p, 3 * self.V, (-1, 3))

    @property
    def tri(self):
        p, t = f64p(), u32p()
        lib.fem_mesh_arrays(self.h, C.byref(p), C.byref(t))
        return self._view(t, 3 * self.F, (-1, 3))

    def topology(self):
        opp, hf, ht, rp, col, rh = i32p(), u32p(), u32p(), i32p(), u32p(), i32p()
        lib.fem_mesh_topology(self.h, C.byref(opp), C.byref(hf), C.byref(ht), C.byref(rp), C.byref(col), C.byref(rh))
        H, V = self.H, self.V
        return dict(opp=self._view(opp, H), he_from=self._view(hf, H), he_to=self._view(ht, H),
                    rowptr=self._view(rp, V + 1), col=self._view(col, H), row_he=self._view(rh, H))

    def init_stats(self, precision=PREC_MIXED):
        _check(lib.fem_mesh_init_stats(self.h, C.c_int(precision)))
        self.precision = precision
        return self

    def init_stats_shard(self, precision, shard):
        """assembly restricted to the rows of a shard's local vertices (fem_mesh_init_stats_shard)"""
        _check(lib.fem_mesh_init_stats_shard(self.h, C.c_int(precision), shard.h))
        self.precision = precision
        return self

    def stats(self):
        ps = [f64p() for _ in range(8)]
        mx = C.c_double()
        _check(lib.fem_mesh_stats(self.h, C.byref(ps[0]), C.byref(ps[1]), C.byref(ps[2]), C.byref(ps[3]),
                                  C.byref(ps[4]), C.byref(ps[5]), C.byref(mx), C.byref(ps[6]), C.byref(ps[7])))
        H, V, F = self.H, self.V, self.F
        return dict(meshAngle=self._view(ps[0], H), meshCot=self._view(ps[1], H), meshArea=self._view(ps[2], F),
                    areaPerVtx=self._view(ps[3], V), wHej=self._view(ps[4], H), wVtx=self._view(ps[5], V),
                    maxEdgeLen=mx.value, gradVec12=self._view(ps[6], 3 * F, (-1, 3)),
                    gradVec13=self._view(ps[7], 3 * F, (-1, 3)))

    def laplacian_cpu(self, fn):
        fn = np.ascontiguousarray(fn, np.float64)
        out = np.empty(self.V)
        _check(lib.fem_compute_laplacian_cpu(self.h, _ptr(fn, f64p), _ptr(out, f64p)))
        return out

    def face_gradients_cpu(self, fn):
        fn = np.ascontiguousarray(fn, np.float64)
        out = np.empty((self.F, 3))
        _check(lib.fem_compute_face_gradients_cpu(self.h, _ptr(fn, f64p), _ptr(out, f64p)))
        return out

    def vertex_gradients_cpu(self, face_grad):
        fg = np.ascontiguousarray(face_grad, np.float64)
        out = np.empty((self.V, 3))
        _check(lib.fem_compute_vertex_gradients_cpu(self.h, _ptr(fg, f64p), _ptr(out, f64p)))
        return out

    def spectrum_bound(self):
        """upper bound on the largest eigenvalue of -L (assembled Laplacian of the stats): the smaller of Gershgorin's and the
        per-triangle bound (fem_mesh_spectrum_bound); the Chebyshev solver of run_imex_fused takes its interval from it"""
        b = C.c_double()
        _check(lib.fem_mesh_spectrum_bound(self.h, C.byref(b)))
        return b.value

    def wide_rows_first(self, width=6):
        """-> (Mesh, new_index): vertices renumbered stably by their number of entries beyond `width`, most first
        (fem_mesh_wide_rows_first); (self, None) when no row is wider"""
        new_index = np.empty(self.V, np.int64)
        oh = vp()
        wide = _check(lib.fem_mesh_wide_rows_first(self.h, C.c_int(width), _ptr(new_index, i64p), C.byref(oh)))
        if wide == 0 or not oh:
            return self, None
        return Mesh(handle=oh), new_index

    def partition_vertices(self, n, method="rcb"):
        """vertex -> part id; method "rcb" (coordinate bisection), "graph" (graph bisection), "auto" (rcb unless its cut says
        the coordinates mislead, then graph) or "multilevel" (multilevel k-way on the dual graph, the METIS_PartMeshDual stage)"""
        npart = np.empty(self.V, np.int64)
        if method == "multilevel":
            cut = C.c_int64()
            _check(lib.fem_partition_vertices_multilevel(self.h, C.c_int(n), _ptr(npart, i64p), None, C.byref(cut)))
            self.last_dual_cut = cut.value
            return npart
        if method == "auto":
            chosen = C.c_int()
            _check(lib.fem_partition_vertices_auto(self.h, C.c_int(n), _ptr(npart, i64p), C.byref(chosen)))
            self.last_partitioner = "graph" if chosen.value == 1 else "rcb"
            return npart
        fn = {"rcb": lib.fem_partition_vertices, "graph": lib.fem_partition_vertices_graph}[method]
        _check(fn(self.h, C.c_int(n), _ptr(npart, i64p)))
        return npart

    def partition(self, n, npart=None, method="rcb"):
        """-> Partition (perm_v, perm_f, node_parts, element_parts, ordered Mesh, cut)"""
        if npart is None and method != "rcb":
            npart = self.partition_vertices(n, method)
        perm_v, perm_f = np.empty(self.V, np.int64), np.empty(self.F, np.int64)
        node_parts, element_parts = np.empty(n + 1, np.uint32), np.empty(n + 2, np.uint32)
        oh = vp()
        if npart is None:
            cut = _check(lib.fem_partition(self.h, C.c_int(n), _ptr(perm_v, i64p), _ptr(perm_f, i64p),
                                           _ptr(node_parts, u32p), _ptr(element_parts, u32p), C.byref(oh)))
        else:
            npart = np.ascontiguousarray(npart, np.int64)
            cut = _check(lib.fem_partition_apply(self.h, C.c_int(n), _ptr(npart, i64p), _ptr(perm_v, i64p),
                                                 _ptr(perm_f, i64p), _ptr(node_parts, u32p),
                                                 _ptr(element_parts, u32p), C.byref(oh)))
        return Partition(n, perm_v, perm_f, node_parts, element_parts, Mesh(handle=oh), cut)


class Partition:
    def __init__(self, n, perm_v, perm_f, node_parts, element_parts, ordered, cut):
        self.n, self.perm_v, self.perm_f = n, perm_v, perm_f
        self.node_parts, self.element_parts, self.ordered, self.cut = node_parts, element_parts, ordered, cut


class RefLayout:
    """Reference device layout (initializeGPUData, reference src/main.cpp:207-342)."""

    def __init__(self, part):
        self.part = part
        h = vp()
        _check(lib.fem_ref_layout_create(part.ordered.h, C.c_int(part.n), _ptr(part.node_parts, u32p),
                                         _ptr(part.element_parts, u32p), C.byref(h)))
        self.h = h
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib.fem_ref_layout_sizes(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        self.maxV, self.W, self.maxF, self.n_halo = a.value, b.value, c.value, d.value

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.fem_ref_layout_destroy(self.h)
            self.h = None

    def arrays(self):
        n, maxV, W, maxF = self.part.n, self.maxV, self.W, self.maxF
        hk, hf, bfc, nbr, fv = u32p(), u32p(), u32p(), u32p(), u32p()
        vw, vvw, fvw = f64p(), f64p(), f64p()
        heg = C.POINTER(C.c_float)()
        lib.fem_ref_layout_arrays(self.h, C.byref(hk), C.byref(hf), C.byref(bfc), C.byref(vw), C.byref(nbr),
                                  C.byref(vvw), C.byref(fv), C.byref(fvw), C.byref(heg))

        def view(p, shape):
            cnt = int(np.prod(shape))
            return np.ctypeslib.as_array(p, shape=(cnt,)).copy().reshape(shape) if cnt else np.zeros(shape)
        return dict(halo_faces_keys=view(hk, (n + 1,)), halo_faces=view(hf, (self.n_halo,)),
                    block_face_count=view(bfc, (n,)), vertex_weights=view(vw, (n, maxV)),
                    nbr_v=view(nbr, (n, W + 1, maxV)), vv_weights=view(vvw, (n, W, maxV)),
                    face_vertices=view(fv, (n, 3, maxF)), fv_weights_new=view(fvw, (n, 3, maxF)),
                    he_grads=view(heg, (n, 2, maxF, 4)))


class RefDevice:
    """Pitched device copies of a RefLayout + two double2 state buffers (GPUrun, main.cpp:344-452)."""

    def __init__(self, layout):
        self.layout = layout
        part = layout.part
        h = vp()
        _check(lib.fem_ref_device_create(layout.h, _ptr(part.node_parts, u32p), C.c_int(part.ordered.V), C.byref(h)))
        self.h, self.V = h, part.ordered.V

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.fem_ref_device_destroy(self.h)
            self.h = None

    def set_state(self, n, c):
        fn = np.ascontiguousarray(np.stack([n, c], axis=1), np.float64)
        _check(lib.fem_ref_device_set_state(self.h, _ptr(fn, f64p)))

    def get_state(self):
        fn = np.empty((self.V, 2))
        _check(lib.fem_ref_device_get_state(self.h, _ptr(fn, f64p)))
        return fn[:, 0].copy(), fn[:, 1].copy()

    def run(self, nsteps, dt, model=None, stream=None):
        mp = C.byref(model) if model is not None else None
        _check(lib.fem_ref_device_run(self.h, C.c_int(nsteps), C.c_double(dt), mp, vp(stream or 0)))

    def args(self):
        """raw device pointers / pitches in step()'s argument order (for direct step() calls)"""
        u64 = [C.c_uint64() for _ in range(10)]
        u32 = [C.c_uint32() for _ in range(10)]
        (fn_one, fn_two, fv, fvw, nbr, vtxW, vvw, grads, parts_n, bfc) = u64
        (fv_p, fvw_p, vw_p, vv_p, vws_p, vv_size, he_p, blocks, threads, smem) = u32
        lib.fem_ref_device_args(self.h, C.byref(fn_one), C.byref(fn_two), C.byref(fv), C.byref(fvw), C.byref(fv_p),
                                C.byref(fvw_p), C.byref(nbr), C.byref(vtxW), C.byref(vw_p), C.byref(vvw),
                                C.byref(vv_p), C.byref(vws_p), C.byref(vv_size), C.byref(grads), C.byref(he_p),
                                C.byref(parts_n), C.byref(bfc), C.byref(blocks), C.byref(threads), C.byref(smem))
        return dict(fn_one=fn_one.value, fn_two=fn_two.value, fv=fv.value, fv_weights=fvw.value,
                    fv_pitch=fv_p.value, fvw_pitch=fvw_p.value, nbr=nbr.value, vtxW=vtxW.value,
                    vw_pitch=vw_p.value, vertex_weights=vvw.value, vv_pitch=vv_p.value, vws_pitch=vws_p.value,
                    vv_size=vv_size.value, grads=grads.value, he_pitch=he_p.value, parts_n=parts_n.value,
                    block_face_count=bfc.value, blocks=blocks.value, threads=threads.value, smem_size=smem.value)


# the reference's own symbol, bound with its 21-argument signature (FEM_common.h:3-25)
lib.step.restype = None
lib.step.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32,
                     C.c_uint32, vp, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32]


def step(a, src, dst, dt):
    """call the reference-signature `step` with the device arrays of RefDevice.args()"""
    lib.step(a[src], a[dst], a["fv"], a["fv_weights"], a["fv_pitch"], a["fvw_pitch"], a["nbr"], a["vtxW"],
             a["vw_pitch"], a["vertex_weights"], a["vv_pitch"], a["vws_pitch"], a["vv_size"], a["grads"],
             a["he_pitch"], a["parts_n"], a["block_face_count"], a["blocks"], a["threads"], dt, a["smem_size"])
    _check(lib.fem_last_error())


class Shard:
    """fem_shard: one rank's contiguous partition range of an ordered mesh + k ghost rings."""

    def __init__(self, part, rank, nranks, krings=1):
        h = vp()
        _check(lib.fem_shard_create(part.ordered.h, C.c_int(part.n), _ptr(part.node_parts, u32p), C.c_int(rank),
                                    C.c_int(nranks), C.c_int(krings), C.byref(h)))
        self.h, self.rank, self.nranks, self.krings = h, rank, nranks, krings
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib.fem_shard_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        self.own_begin, self.own_end, self.n_local, self.n_send = a.value, b.value, c.value, d.value
        self.n_own = self.own_end - self.own_begin
        l2g, sidx, sc, rc = u32p(), u32p(), C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
        lib.fem_shard_arrays(self.h, C.byref(l2g), C.byref(sidx), C.byref(sc), C.byref(rc))
        self.loc2glob = np.ctypeslib.as_array(l2g, shape=(self.n_local,)).copy()
        self.send_idx = (np.ctypeslib.as_array(sidx, shape=(self.n_send,)).copy() if self.n_send
                         else np.zeros(0, np.uint32))
        self.send_counts = np.ctypeslib.as_array(sc, shape=(nranks,)).copy()
        self.recv_counts = np.ctypeslib.as_array(rc, shape=(nranks,)).copy()

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.fem_shard_destroy(self.h)
            self.h = None


class Plan:
    """Native plan: sliced-ELL operator + ping-pong state on the device."""

    def __init__(self, ordered, precision=PREC_F32, model=None, shard=None, part=None, ksteps=0, assemble=False,
                 implicit=False, mixed=False):
        """ksteps > 0 (with part = the Partition of `ordered`): fused-tile layout, up to ksteps time
        steps per launch; otherwise the sliced-ELL layout, one step per launch.
        assemble=True keeps connectivity/positions on the device for fem_plan_assemble;
        implicit=True (sliced-ELL plan of a whole mesh) enables run_imex;
        mixed=True (fused fp64 plans): the reference's own precision mix -- fp64 state / Laplacian / update, fp32 gradient
        branch (FEM_PLAN_MIXED_RECORDS)."""
        h = vp()
        mp = C.byref(model) if model is not None else None
        self.ksteps = ksteps
        self.F = ordered.F
        if assemble or implicit or mixed:
            _check(lib.fem_plan_create_ex(ordered.h, C.c_int(part.n if part is not None else 0),
                                          _ptr(part.node_parts, u32p) if part is not None else None,
                                          shard.h if shard is not None else None, C.c_int(precision), mp,
                                          C.c_int(ksteps), C.c_uint((1 if assemble else 0) | (2 if implicit else 0) | (4 if mixed else 0)), C.byref(h)))
            self.V = shard.n_local if shard is not None else ordered.V
        elif ksteps > 0:
            _check(lib.fem_plan_create_tiled(ordered.h, C.c_int(part.n), _ptr(part.node_parts, u32p),
                                             shard.h if shard is not None else None, C.c_int(precision), mp,
                                             C.c_int(ksteps), C.byref(h)))
            self.V = shard.n_local if shard is not None else ordered.V
        elif shard is None:
            _check(lib.fem_plan_create(ordered.h, C.c_int(precision), mp, C.byref(h)))
            self.V = ordered.V
        else:
            _check(lib.fem_plan_create_shard(ordered.h, shard.h, C.c_int(precision), mp, C.byref(h)))
            self.V = shard.n_local
        self.h, self.precision = h, precision

    def set_points(self, points):
        pts = np.ascontiguousarray(points, np.float64)
        _check(lib.fem_plan_set_points(self.h, _ptr(pts, f64p)))

    def deform(self, amplitude, wavenumber, phase, stream=None):
        _check(lib.fem_plan_deform(self.h, C.c_double(amplitude), C.c_double(wavenumber), C.c_double(phase),
                                   vp(stream or 0)))

    def assemble(self, stream=None):
        _check(lib.fem_plan_assemble(self.h, vp(stream or 0)))

    def assemble_step(self, dt, model=None, store_operator=False, stream=None):
        """assembly from the current positions fused with one time step (fem_plan_assemble_step)"""
        mp = C.byref(model) if model is not None else None
        _check(lib.fem_plan_assemble_step(self.h, C.c_double(dt), mp, C.c_int(1 if store_operator else 0), vp(stream or 0)))

    def max_edge(self):
        v = C.c_double()
        _check(lib.fem_plan_max_edge(self.h, C.byref(v)))
        return v.value

    def face_records(self):
        out = np.empty((self.F, 16))
        _check(lib.fem_plan_face_records(self.h, _ptr(out, f64p)))
        return out

    def coef(self):
        out = np.empty(lib.fem_plan_coef_count(self.h))
        _check(lib.fem_plan_download_coef(self.h, _ptr(out, f64p)))
        return out.reshape(-1, 4)

    def format(self, nverts=None):
        """-> (rgba float64 [V,4], rgb uint8 [V,3], nmin, nmax): the reference's colour map of n"""
        V = nverts or self.V
        rgba, rgb = np.empty((V, 4)), np.empty((V, 3), np.uint8)
        lo, hi = C.c_double(), C.c_double()
        _check(lib.fem_plan_format(self.h, C.c_uint32(V), _ptr(rgba, f64p), _ptr(rgb, u8p), C.byref(lo), C.byref(hi)))
        return rgba, rgb, lo.value, hi.value

    def format_noalloc(self, dev_rgba, dev_minmax=0, nverts=None, stream=None):
        """fem_plan_format_noalloc: colours into caller-owned device memory, no allocation, async on stream"""
        _check(lib.fem_plan_format_noalloc(self.h, C.c_uint32(nverts or self.V), C.c_uint64(dev_rgba), C.c_uint64(dev_minmax),
                                           vp(stream or 0)))

    def tile_info(self):
        k, thr, rpt = C.c_int(), C.c_int(), C.c_int()
        nt, ml, rows = C.c_uint32(), C.c_uint32(), C.c_uint64()
        _check(lib.fem_plan_tile_info(self.h, C.byref(k), C.byref(nt), C.byref(ml), C.byref(rows), C.byref(thr),
                                      C.byref(rpt)))
        return dict(ksteps=k.value, ntiles=nt.value, max_local=ml.value, rows_loaded=rows.value,
                    threads=thr.value, rows_per_thread=rpt.value, groups_per_launch=max(1, self.persist_groups()),
                    operator_records={1: "12-byte (fp32 weight + 3 x 21-bit gradient coefficients, per-row scale)",
                                      2: "20-byte (fp64 weight + 3 x fp32 gradient coefficients)"}.get(self.operator_format(), "4 scalars"))

    def pack_halo(self, dst_ptr, stream=None, which=0):
        """which = 0: from the current state buffer, 1: from the next one (see launch_tiles)"""
        _check(lib.fem_plan_pack_halo_from(self.h, which, dst_ptr, vp(stream or 0)))

    def persist_groups(self, with_min=False):
        """groups of ksteps steps one launch of run() may cover: 0 = one launch per group (tile_step_kernel), n > 0 = up
        to n groups per launch of tile_step_persist_kernel (fem_plan_persist_groups); with_min: (n, the fewest groups
        a persistent launch is used for -- a run's remainder below that goes one launch per group)"""
        g, gmin = C.c_int(), C.c_int()
        _check(lib.fem_plan_persist_groups(self.h, C.byref(g), C.byref(gmin)))
        return (g.value, gmin.value) if with_min else g.value

    def launch_count(self, nsteps):
        """(launches, persistent launches among them) run(nsteps) makes on a fused plan"""
        k = self.tile_info()["ksteps"]
        g, gmin = self.persist_groups(with_min=True)
        total, persistent, left = 0, 0, int(nsteps)
        while left > 0:
            groups = min(g, -(-left // k))
            if groups < max(gmin, 1):
                total += -(-left // k)
                break
            total, persistent, left = total + 1, persistent + 1, left - min(left, groups * k)
        return total, persistent

    def tile_classes(self):
        """(interior, boundary, ghost) tile counts; the tiles are ordered in that sequence"""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib.fem_plan_tile_classes(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def launch_tiles(self, first, count, nsteps, dt, model=None, stream=None):
        """nsteps <= ksteps fused steps of tiles [first, first+count): current buffer -> next; no flip"""
        mp = C.cast(C.byref(model), vp) if model is not None else None
        _check(lib.fem_plan_launch_tiles(self.h, first, count, nsteps, dt, mp, vp(stream or 0)))

    def flip(self):
        _check(lib.fem_plan_flip(self.h))

    def state_buffer(self, which=0):
        p = C.c_uint64()
        _check(lib.fem_plan_state_buffer(self.h, which, C.byref(p)))
        return p.value

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.fem_plan_destroy(self.h)
            self.h = None

    def set_state(self, n, c):
        n, c = np.ascontiguousarray(n, np.float64), np.ascontiguousarray(c, np.float64)
        _check(lib.fem_plan_set_state(self.h, _ptr(n, f64p), _ptr(c, f64p)))

    def get_state(self):
        n, c = np.empty(self.V), np.empty(self.V)
        _check(lib.fem_plan_get_state(self.h, _ptr(n, f64p), _ptr(c, f64p)))
        return n, c

    def run(self, nsteps, dt, model=None, stream=None):
        mp = C.byref(model) if model is not None else None
        _check(lib.fem_plan_run(self.h, C.c_int(nsteps), C.c_double(dt), mp, vp(stream or 0)))

    def run_imex(self, nsteps, dt, model=None, cg_iterations=20, stream=None):
        """semi-implicit steps: diffusion implicit (preconditioned CG), reaction/chemotaxis explicit"""
        mp = C.byref(model) if model is not None else None
        _check(lib.fem_plan_run_imex(self.h, C.c_int(nsteps), C.c_double(dt), mp, C.c_int(cg_iterations), vp(stream or 0)))

    def run_imex_fused(self, nsteps, dt, model=None, rel_tol=1e-6, stream=None):
        """semi-implicit steps on a fused-tile plan: Chebyshev iteration, ksteps iterations per launch (fem_plan_run_imex_fused)"""
        mp = C.byref(model) if model is not None else None
        _check(lib.fem_plan_run_imex_fused(self.h, C.c_int(nsteps), C.c_double(dt), mp, C.c_double(rel_tol), vp(stream or 0)))

    def imex_fused_iterations(self):
        return int(lib.fem_plan_imex_fused_iterations(self.h))

    def imex_tolerance(self, rel_tol):
        """residual-based stop of the CG solves (0 = fixed iteration count)"""
        _check(lib.fem_plan_imex_tolerance(self.h, C.c_double(rel_tol)))

    def imex_iterations(self):
        it = C.c_int()
        _check(lib.fem_plan_imex_iterations(self.h, C.byref(it)))
        return it.value

    def imex_residual(self):
        a, b = C.c_double(), C.c_double()
        _check(lib.fem_plan_imex_residual(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def operator_bytes(self):
        return lib.fem_plan_operator_bytes(self.h)

    def operator_format(self):
        """0: four scalars per entry; 1: 12-byte records (fem_plan_operator_format)"""
        return lib.fem_plan_operator_format(self.h)

    def state_ptr(self):
        p, nv, sb, bn, bc = C.c_uint64(), C.c_uint64(), C.c_int(), C.c_double(), C.c_double()
        _check(lib.fem_plan_state(self.h, C.byref(p), C.byref(nv), C.byref(sb), C.byref(bn), C.byref(bc)))
        return p.value, nv.value, sb.value, bn.value, bc.value
