"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.

Bars:
  * `step` / fem_step on the reference layout (fp64 state, fp32 gradient branch): rel-L2 <= 3e-8
    (TOL_REF below says why not 1e-8) against the oracle's mixed-precision CPUrun.  Not bit-exact
    by construction: the fp32 products of the gradient branch are contracted differently by
    different compilers, and the reference's own float-atomic scatter is order-dependent (error
    per step ~ alpha*dt*|Gn.Gc|*2^-24, about 1e-10 absolute on these inputs).  The reference holds
    no fixture at this level, so this bar is pinned by this repository's oracle alone.
  * native plan, fp64: rel-L2 <= 1e-11 against the all-fp64 oracle (same arithmetic, folded).
  * native plan, fp32: rel-L2 <= 1e-5 against the all-fp64 oracle after 100 steps from a
    smooth initial condition (the tolerance BASELINE.json's north_star states).
Every comparison is made twice (helpers.parity): relative to the fields, as the north star words it,
and relative to the DEVIATION of the reference from the homogeneous state (||n - nMax|| ~ 0.003 ||n||
here) -- fp32 and the drop-in step <= 1e-5 of it, fp64 <= 1e-9 -- so that an error of the pattern
itself cannot hide behind the magnitude of n.
"""
import numpy as np
import pytest

import fem_amd as fa
import oracle_lib as ol
from helpers import parity, rel_l2, scan_case, smooth_ic, sphere_case, stable_dt, strip_case

pytestmark = pytest.mark.gpu

# drop-in step() against the oracle's MIXED mode: the reference keeps the half-edge gradients and their products in fp32
# (float4 he_grads, FEM_common.cu:104-132), so the two sides agree only up to how the compiler contracts those fp32
# multiply-adds (nvcc and hipcc both do by default; the oracle is C without contraction): observed 2e-10 .. 1.4e-8
# (the largest on the single-triangle mesh, where nothing averages out).  Measured in round 3: with `#pragma clang fp
# contract(off)` inside ref_step_kernel the 100-step figures on the nu=32 sphere / the strip are 1.26e-11 / 7.0e-12 against
# 1.43e-11 / 1.41e-11 with the default contraction, at the same speed -- what is left is the order of the fp32 corner sums,
# so the pragma was not adopted (it would only pin the kernel's bits against future flag changes).
TOL_REF = 3e-8
TOL_F64 = 1e-11
TOL_F32 = 1e-5
# fp32 plans, error relative to the DEVIATION ||n_ref - nMax|| (helpers.parity's second bar; default 1e-5 of it, which the
# rough random initial condition of main.cpp:182-186 meets: bench.py's 25 steps at 16.4M give 1.1e-7).  From a SMOOTH
# initial condition fp32 cannot hold that: n's deviation decays to a quasi-equilibrium r*dn ~ -alpha*lap(dc), and the
# Laplacian of a smooth fp32 field is a second difference of nearly equal numbers (w ~ 1/h^2, c_j - c_i ~ 1e-6 at a
# rounding of 7e-11), so the error of n is ~1e-3 of the (small) deviation of n while c stays at ~5e-6 of its own.  The
# bars below are ~5x the observed values (MI355X, round 3), kept next to each call; a regression of one order shows.
DEV_SMOOTH_100 = 1e-3        # observed 1.4e-4 (sphere nu=32), 6.0e-5 (strip), 5.5e-4 (nu=64) after 100 steps
DEV_SMOOTH_LARGE = 1e-2      # observed 2.1e-3 (1M, 20 steps), 1.6e-3 (16.4M, 12 steps), 1.8e-3 (irregular, 100 steps)


def _ref_device(case):
    case.mesh.init_stats(fa.PREC_MIXED)
    lay = fa.RefLayout(case.part)
    return lay, fa.RefDevice(lay)


@pytest.mark.parametrize("make,nparts", [(lambda n: sphere_case(8, n), 4), (lambda n: strip_case(n), 3),
                                         (lambda n: sphere_case(32, n), 52)])
def test_reference_step_symbol(make, nparts):
    """the reference's own 21-argument `step`, ping-ponged as at main.cpp:481-483"""
    case = make(nparts)
    lay, dev = _ref_device(case)
    dev.set_state(case.n0, case.c0)
    a = dev.args()
    fa.step(a, "fn_one", "fn_two", case.dt)
    fa.step(a, "fn_two", "fn_one", case.dt)
    n_gpu, c_gpu = dev.get_state()          # fn_one is current again
    n_ref, c_ref = case.ostats(ol.MIXED).step(case.n0, case.c0, case.dt, 2, case.model.as_array())
    parity(n_gpu, c_gpu, n_ref, c_ref, TOL_REF)
    assert not np.array_equal(n_gpu, case.n0)


def test_reference_step_is_bit_reproducible():
    """the corner sums run in a fixed order (no floating-point atomics): same input, same bits"""
    case = sphere_case(24, 16, seed=5)
    outs = []
    for _ in range(3):
        lay, dev = _ref_device(case)
        dev.set_state(case.n0, case.c0)
        dev.run(7, case.dt, case.model)
        outs.append(dev.get_state())
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])


def test_reference_layout_run_vs_packed_oracle():
    """fem_ref_device_run (20 steps) against the literal CPU model of stepKernel on the oracle's
    own packed arrays"""
    case = sphere_case(8, 5)
    lay, dev = _ref_device(case)
    dev.set_state(case.n0, case.c0)
    dev.run(20, case.dt, case.model)
    n_gpu, c_gpu = dev.get_state()
    olay = ol.pack_ref_layout(case.ostats(ol.MIXED), case.n, case.part.node_parts, case.part.element_parts)
    fn = np.stack([case.n0, case.c0], axis=1)
    for _ in range(20):
        fn = ol.step_packed(olay, case.n, case.part.node_parts, case.model.as_array(), case.dt, fn)
    parity(n_gpu, c_gpu, fn[:, 0], fn[:, 1], TOL_REF)


def test_reference_step_custom_model_and_clamp():
    """alt constant set of main.cpp:49-54 and the max(0, .) clamp (FEM_common.cu:164-165).
    Rough state + large dt: the fp32 gradient branch dominates the error, hence the looser bar."""
    case = strip_case(4)
    lay, dev = _ref_device(case)
    model = fa.default_model()
    model.r, model.alpha = 1.52, 12.02
    n0 = case.n0.copy()
    c0 = case.c0.copy()
    n0[::7] = 0.0                       # neighbours of these are driven negative -> clamped to 0
    dev.set_state(n0, c0)
    dev.run(2, 20 * case.dt, model)
    n_gpu, c_gpu = dev.get_state()
    n_ref, c_ref = case.ostats(ol.MIXED).step(n0, c0, 20 * case.dt, 2, model.as_array())
    assert (n_ref == 0).sum() > 10
    assert (n_gpu >= 0).all() and (c_gpu >= 0).all()
    assert np.abs(n_gpu[n_ref == 0]).max() <= 1e-5 and np.abs(n_ref[n_gpu == 0]).max() <= 1e-5
    parity(n_gpu, c_gpu, n_ref, c_ref, 1e-6)


@pytest.mark.parametrize("ksteps", [0, 3])
def test_native_plan_clamp_and_custom_model(ksteps):
    """the max(0, .) clamp and the alt constant set on the native plans: in the deviation-form
    state a clamped value is stored as -base, which reads back as exactly 0"""
    case = strip_case(4)
    case.mesh.init_stats(fa.PREC_F64)
    model = fa.default_model()
    model.r, model.alpha = 1.52, 12.02
    n0, c0 = case.n0.copy(), case.c0.copy()
    n0[::7] = 0.0
    dt = 20 * case.dt
    n_ref, c_ref = case.ostats(ol.F64).step(n0, c0, dt, 2, model.as_array())
    assert (n_ref == 0).sum() > 10
    plan = fa.Plan(case.mesh, fa.PREC_F64, model, part=case.part, ksteps=ksteps)
    plan.set_state(n0, c0)
    plan.run(2, dt, model)
    n, c = plan.get_state()
    assert np.array_equal(n == 0, n_ref == 0) and (n >= 0).all() and (c >= 0).all()
    parity(n, c, n_ref, c_ref, TOL_F64)
    # non-default nMax: the base state the deviations are taken from follows the model
    model.nMax = 1.7
    plan2 = fa.Plan(case.mesh, fa.PREC_F32, model, part=case.part, ksteps=ksteps)
    nb, cb = 1.7 - (case.n0 - 1.0), 1.0 / (1.0 + 1.7 - (case.n0 - 1.0))
    plan2.set_state(nb, cb)
    plan2.run(5, case.dt, model)
    n2, c2 = plan2.get_state()
    r2 = case.ostats(ol.F64).step(nb.astype(np.float32).astype(np.float64), cb.astype(np.float32).astype(np.float64), case.dt, 5,
                                  model.as_array())
    # c0 = 1/(1+n) is far from the base state nMax/(1+nMax) here, so c is stored as a LARGE fp32 deviation (-0.26) and
    # its rounding feeds n through alpha*lap(c): observed 1.1e-4 of n's deviation
    parity(n2, c2, r2[0], r2[1], TOL_F32, nMax=1.7, tol_dev=1e-3)


def test_reference_step_limits_and_errors():
    case = sphere_case(4, 1)            # one partition, 162 vertices, 320 faces
    lay, dev = _ref_device(case)
    a = dev.args()
    # aliased src/dst is rejected, nothing launched, error retrievable as the void shim requires
    fa.lib.step(a["fn_one"], a["fn_one"], a["fv"], a["fv_weights"], a["fv_pitch"], a["fvw_pitch"], a["nbr"],
                a["vtxW"], a["vw_pitch"], a["vertex_weights"], a["vv_pitch"], a["vws_pitch"], a["vv_size"],
                a["grads"], a["he_pitch"], a["parts_n"], a["block_face_count"], a["blocks"], a["threads"], 0.001,
                a["smem_size"])
    assert fa.lib.fem_last_error() == -1
    # > 1024 vertices per partition: refused like GPUrun does (main.cpp:357-360)
    big = sphere_case(12, 1)
    big.mesh.init_stats(fa.PREC_MIXED)
    with pytest.raises(fa.FemError):
        fa.RefDevice(fa.RefLayout(big.part))


@pytest.mark.parametrize("make,nparts,steps", [(lambda n: sphere_case(8, n), 4, 10), (lambda n: strip_case(n), 3, 10),
                                               (lambda n: sphere_case(32, n), 52, 100)])
def test_native_plan_f64(make, nparts, steps):
    case = make(nparts)
    case.mesh.init_stats(fa.PREC_F64)
    plan = fa.Plan(case.mesh, fa.PREC_F64)
    plan.set_state(case.n0, case.c0)
    plan.run(steps, case.dt, case.model)
    n_gpu, c_gpu = plan.get_state()
    n_ref, c_ref = case.ostats(ol.F64).step(case.n0, case.c0, case.dt, steps, case.model.as_array())
    parity(n_gpu, c_gpu, n_ref, c_ref, TOL_F64)


@pytest.mark.parametrize("make,nparts", [(lambda n: sphere_case(32, n), 52), (lambda n: strip_case(n), 3)])
def test_native_plan_f32_100_steps(make, nparts):
    """BASELINE config 1 shape: V = 10 242 sphere (P = 52) and the 451-vertex strip, 100 steps"""
    case = make(nparts)
    case.mesh.init_stats(fa.PREC_F64)
    n0, c0 = smooth_ic(case.pts)
    plan = fa.Plan(case.mesh, fa.PREC_F32)
    plan.set_state(n0, c0)
    plan.run(100, case.dt, case.model)
    n_gpu, c_gpu = plan.get_state()
    # the fp32 path starts from the fp32-rounded state; give the oracle the same start
    n0r, c0r = n0.astype(np.float32).astype(np.float64), c0.astype(np.float32).astype(np.float64)
    n_ref, c_ref = case.ostats(ol.F64).step(n0r, c0r, case.dt, 100, case.model.as_array())
    parity(n_gpu, c_gpu, n_ref, c_ref, TOL_F32, tol_dev=DEV_SMOOTH_100)


def test_native_plan_f32_random_ic_and_mixed_oracle():
    """random (rough) IC of main.cpp:182-186, 100 steps; also within tolerance of the reference's
    own precision mix"""
    case = sphere_case(32, 52, seed=7)
    case.mesh.init_stats(fa.PREC_F64)
    plan = fa.Plan(case.mesh, fa.PREC_F32)
    plan.set_state(case.n0, case.c0)
    plan.run(100, case.dt, case.model)
    n_gpu, c_gpu = plan.get_state()
    n0r, c0r = case.n0.astype(np.float32).astype(np.float64), case.c0.astype(np.float32).astype(np.float64)
    for prec in (ol.F64, ol.MIXED):
        n_ref, c_ref = case.ostats(prec).step(n0r, c0r, case.dt, 100, case.model.as_array())
        # 100 steps: the rough part of the random IC has diffused away; observed 1.8e-5 of the deviation (n and c)
        parity(n_gpu, c_gpu, n_ref, c_ref, TOL_F32, tol_dev=1e-4)


def test_native_plan_ragged_sizes():
    """V not a multiple of 64, a single slice, boundary vertices, valence-5 rows"""
    for pts_tri, nparts in ((fa.make_icosphere(1, 0.1), 1), (fa.make_icosphere(3, 0.1), 2), (fa.make_strip(2, 2, 0.1), 1),
                            (fa.make_strip(5, 13, 0.1), 2)):
        from helpers import Case
        case = Case(pts_tri[0], pts_tri[1], nparts)
        case.mesh.init_stats(fa.PREC_F64)
        plan = fa.Plan(case.mesh, fa.PREC_F64)
        plan.set_state(case.n0, case.c0)
        plan.run(5, case.dt, case.model)
        n_gpu, c_gpu = plan.get_state()
        n_ref, c_ref = case.ostats(ol.F64).step(case.n0, case.c0, case.dt, 5, case.model.as_array())
        parity(n_gpu, c_gpu, n_ref, c_ref, TOL_F64)


@pytest.mark.parametrize("ksteps", [0, 4])
def test_native_plan_full_size_properties(ksteps):
    """1M-vertex sphere (BASELINE config 2 size), sliced-ELL and fused plans: size-independent
    properties.
    (i) a constant state is a fixed point of diffusion/chemotaxis: with r = 0 one step changes
        n by exactly 0 and c only through S*(n/(1+n) - c) = 0;
    (ii) determinism: two runs from the same state agree bit for bit (no atomics);
    (iii) the fp32 result matches the fp64 plan within tolerance;
    (iv) format(): extrema and colours of the result agree with numpy on the downloaded state."""
    pts, tri = fa.make_icosphere(320, 0.1)
    mesh = fa.Mesh(pts, tri)
    part = mesh.partition(4096 if ksteps == 0 else mesh.V // 620)
    om = part.ordered
    om.init_stats(fa.PREC_F64)
    dt = 0.1 * om.stats()["maxEdgeLen"] ** 2
    model = fa.default_model()
    plan = fa.Plan(om, fa.PREC_F32, part=part, ksteps=ksteps)
    # (i)
    m0 = fa.default_model()
    m0.r = 0.0
    n_const = np.full(om.V, 0.75)
    c_const = n_const / (1 + n_const)
    plan.set_state(n_const, c_const)
    plan.run(1, dt, m0)
    n1, c1 = plan.get_state()
    assert np.array_equal(n1, n_const.astype(np.float32).astype(np.float64))
    assert np.abs(c1 - c_const).max() <= 1e-7
    # (ii) + (iii)
    n0, c0 = smooth_ic(om.points)
    outs = []
    for _ in range(2):
        plan.set_state(n0, c0)
        plan.run(20, dt, model)
        outs.append(plan.get_state())
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    plan64 = fa.Plan(om, fa.PREC_F64, part=part, ksteps=ksteps)
    plan64.set_state(n0.astype(np.float32), c0.astype(np.float32))
    plan64.run(20, dt, model)
    n64, c64 = plan64.get_state()
    parity(outs[0][0], outs[0][1], n64, c64, TOL_F32, tol_dev=DEV_SMOOTH_LARGE)
    # (iv)
    rgba, rgb, lo, hi = plan.format()
    n = outs[0][0]
    assert abs(lo - n.min()) <= 1e-6 and abs(hi - n.max()) <= 1e-6
    assert np.abs(rgba[:, 0] - (n - n.min()) / (n.max() - n.min())).max() <= 1e-4


@pytest.mark.parametrize("make,nparts,ksteps,steps", [
    (lambda n: sphere_case(32, n), 52, 4, 10),        # 10 = 4 + 4 + 2: a short last launch
    (lambda n: sphere_case(32, n), 8, 3, 9),          # partitions of 1280 vertices: split into tiles
    (lambda n: strip_case(n), 3, 8, 16),              # boundary vertices, rings that run out
    (lambda n: sphere_case(3, n), 2, 2, 5),           # tiny: rings cover the whole mesh
    (lambda n: sphere_case(40, n), 16, 1, 3),         # ksteps = 1
])
def test_fused_tile_plan_f64(make, nparts, ksteps, steps):
    """fused k-step kernel against the all-fp64 oracle (same arithmetic as the single-step plan)"""
    case = make(nparts)
    case.mesh.init_stats(fa.PREC_F64)
    plan = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps)
    plan.set_state(case.n0, case.c0)
    plan.run(steps, case.dt, case.model)
    n_gpu, c_gpu = plan.get_state()
    n_ref, c_ref = case.ostats(ol.F64).step(case.n0, case.c0, case.dt, steps, case.model.as_array())
    parity(n_gpu, c_gpu, n_ref, c_ref, TOL_F64)


@pytest.mark.parametrize("cfg", [0, 2, 6, 7, 8])
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_every_register_tile_geometry(cfg, prec, monkeypatch):
    """FEM_TILE_CONFIG: 512 x 2 / x 3 / x 4 register rows (the defaults by mesh), 512 x 2 at three workgroups
    per CU, 768 x 2 -- on a regular mesh (partitions of 1000 rows: split for the small tiles) and on the
    irregular one (wide rows: LDS extras), several launches with a short last one"""
    monkeypatch.setenv("FEM_TILE_CONFIG", str(cfg))
    for case, steps, ksteps in ((sphere_case(40, 16), 11, 4), (scan_case(24, 8), 9, 4)):
        case.mesh.init_stats(fa.PREC_F64)
        st = case.ostats(ol.F64)
        dt = min(case.dt, stable_dt(st, case.model))
        plan = fa.Plan(case.mesh, fa.PREC_F64 if prec == "f64" else fa.PREC_F32, part=case.part, ksteps=ksteps)
        assert plan.tile_info()["threads"] * plan.tile_info()["rows_per_thread"] == {0: 1024, 2: 1536, 6: 1024, 7: 1536, 8: 2048}[cfg]
        n0 = case.n0.astype(np.float32).astype(np.float64) if prec == "f32" else case.n0
        c0 = case.c0.astype(np.float32).astype(np.float64) if prec == "f32" else case.c0
        plan.set_state(n0, c0)
        plan.run(steps, dt, case.model)
        n_gpu, c_gpu = plan.get_state()
        n_ref, c_ref = st.step(n0, c0, dt, steps, case.model.as_array())
        tol = TOL_F64 if prec == "f64" else 1e-5
        parity(n_gpu, c_gpu, n_ref, c_ref, tol)


def test_fused_tile_plan_f32_matches_single_step_plan():
    """fp32: the fused kernel and the single-step kernel do the same arithmetic per row"""
    case = sphere_case(64, 40)
    case.mesh.init_stats(fa.PREC_F64)
    n0, c0 = smooth_ic(case.pts)
    fused = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=4)
    single = fa.Plan(case.mesh, fa.PREC_F32)
    outs = []
    for plan in (fused, single):
        plan.set_state(n0, c0)
        plan.run(100, case.dt, case.model)
        outs.append(plan.get_state())
    parity(outs[0][0], outs[0][1], outs[1][0], outs[1][1], 1e-6)
    n0r, c0r = n0.astype(np.float32).astype(np.float64), c0.astype(np.float32).astype(np.float64)
    n_ref, c_ref = case.ostats(ol.F64).step(n0r, c0r, case.dt, 100, case.model.as_array())
    parity(outs[0][0], outs[0][1], n_ref, c_ref, TOL_F32, tol_dev=5 * DEV_SMOOTH_100)


def test_irregular_mesh_all_paths():
    """BASELINE config 4 shape (irregular valences 3..16, thin triangles, negative cotangent
    weights): the drop-in `step`, the sliced-ELL plan and the fused plan (rows wider than the
    register tile go through the LDS overflow entries) against the oracle"""
    case = scan_case(24, 16)                          # <= 1024 faces per partition (reference limit)
    deg = np.diff(case.topo.rowptr)
    assert deg.max() >= 12 and deg.min() <= 4
    st = case.ostats(ol.F64)
    dt = stable_dt(st, case.model)
    n0, c0 = smooth_ic(case.pts)
    marr = case.model.as_array()
    # reference symbol on the reference layout (mixed precision)
    case.mesh.init_stats(fa.PREC_MIXED)
    lay, dev = _ref_device(case)
    dev.set_state(n0, c0)
    dev.run(10, dt, case.model)
    n_gpu, c_gpu = dev.get_state()
    n_ref, c_ref = case.ostats(ol.MIXED).step(n0, c0, dt, 10, marr)
    parity(n_gpu, c_gpu, n_ref, c_ref, TOL_REF)
    # native plans, fp64
    case.mesh.init_stats(fa.PREC_F64)
    n_ref, c_ref = st.step(n0, c0, dt, 22, marr)
    outs = {}
    for ksteps in (0, 1, 4):
        plan = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps)
        plan.set_state(n0, c0)
        plan.run(22, dt, case.model)
        outs[ksteps] = plan.get_state()
        parity(outs[ksteps][0], outs[ksteps][1], n_ref, c_ref, TOL_F64)
    # same accumulation order in both kernels: bit-identical results
    assert np.array_equal(outs[0][0], outs[4][0]) and np.array_equal(outs[0][1], outs[4][1])
    # fp32
    n0r, c0r = n0.astype(np.float32).astype(np.float64), c0.astype(np.float32).astype(np.float64)
    n_ref, c_ref = st.step(n0r, c0r, dt, 100, marr)
    for ksteps in (0, 4):
        plan = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=ksteps)
        plan.set_state(n0, c0)
        plan.run(100, dt, case.model)
        n_gpu, c_gpu = plan.get_state()
        parity(n_gpu, c_gpu, n_ref, c_ref, TOL_F32, tol_dev=DEV_SMOOTH_LARGE)


def test_irregular_mesh_small_tiles_and_sharded_plan():
    """many overflow entries per tile (5 % of the vertices fanned to valence 16), partitions that
    must be split, and a shard's plan (ghost rows in their own tiles)"""
    from helpers import Case
    pts, tri = fa.make_scanned_like(20, 0.1, seed=9, flip_fraction=0.3, high_valence_fraction=0.05)
    case = Case(pts, tri, 3)
    case.mesh.init_stats(fa.PREC_F64)
    st = case.ostats(ol.F64)
    dt = stable_dt(st, case.model)
    n_ref, c_ref = st.step(case.n0, case.c0, dt, 9, case.model.as_array())
    plan = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=3)
    plan.set_state(case.n0, case.c0)
    plan.run(9, dt, case.model)
    n_gpu, c_gpu = plan.get_state()
    parity(n_gpu, c_gpu, n_ref, c_ref, TOL_F64)
    # shard 1 of 3 with 3 ghost rings: 3 steps without exchange are exact on the owned rows
    shard = fa.Shard(case.part, 1, 3, 3)
    l2g = shard.loc2glob
    n3, c3 = st.step(case.n0, case.c0, dt, 3, case.model.as_array())
    for ksteps in (0, 3):
        sp = fa.Plan(case.mesh, fa.PREC_F64, shard=shard, part=case.part, ksteps=ksteps)
        sp.set_state(case.n0[l2g], case.c0[l2g])
        sp.run(3, dt, case.model)
        ns, cs = sp.get_state()
        own = slice(0, shard.n_own)
        parity(ns[own], cs[own], n3[l2g[own]], c3[l2g[own]], TOL_F64)


def test_config3_full_size_properties():
    """BASELINE config 3 size (nu = 1280, V = 16 384 002), fp32: the fused kernel and the
    single-step kernel -- two layouts, two kernels -- agree after 12 steps; a constant state is a
    fixed point; a second run reproduces the first bit for bit; the fp64 plan is within 1e-5."""
    pts, tri = fa.make_icosphere(1280, 0.1)
    mesh = fa.Mesh(pts, tri)
    del tri
    part = mesh.partition(mesh.V // 620)
    del mesh
    om = part.ordered
    om.init_stats(fa.PREC_F64)
    dt = 0.1 * om.stats()["maxEdgeLen"] ** 2
    model = fa.default_model()
    n0, c0 = smooth_ic(om.points)
    fused = fa.Plan(om, fa.PREC_F32, part=part, ksteps=4)
    fused.set_state(n0, c0)
    fused.run(12, dt, model)
    a = fused.get_state()
    fused.set_state(n0, c0)
    fused.run(12, dt, model)
    b = fused.get_state()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    m0 = fa.default_model()
    m0.r = 0.0
    n_const = np.full(om.V, 0.75)
    fused.set_state(n_const, n_const / (1 + n_const))
    fused.run(4, dt, m0)
    n1, c1 = fused.get_state()
    assert np.array_equal(n1, n_const.astype(np.float32).astype(np.float64)) and np.abs(c1 - n_const / (1 + n_const)).max() <= 1e-7
    del fused
    single = fa.Plan(om, fa.PREC_F32)
    single.set_state(n0, c0)
    single.run(12, dt, model)
    s = single.get_state()
    parity(a[0], a[1], s[0], s[1], 1e-6)
    assert np.isfinite(s[0]).all() and s[0].min() >= 0 and s[1].min() >= 0
    del single
    # and both are within the stated fp32 tolerance of the all-fp64 plan at this size
    p64 = fa.Plan(om, fa.PREC_F64)
    p64.set_state(n0.astype(np.float32), c0.astype(np.float32))
    p64.run(12, dt, model)
    d = p64.get_state()
    parity(a[0], a[1], d[0], d[1], TOL_F32, tol_dev=DEV_SMOOTH_LARGE)


def test_edge_cases_isolated_vertices_and_tiny_meshes():
    """vertices without faces ("0 if no incident face", mesh_function.h:61-110), a single triangle,
    one partition, more fused steps than the mesh has rings: every path against the oracle"""
    from helpers import Case
    pts, tri = fa.make_icosphere(4, 0.1)
    extra = np.array([[0.3, 0.1, -0.2], [5.0, 5.0, 5.0], [-1.0, 0.2, 0.4]])
    pts2 = np.vstack([pts[:50], extra[:1], pts[50:], extra[1:]])          # isolated vertices in the middle and at the end
    remap = np.concatenate([np.arange(50), np.arange(51, 51 + len(pts) - 50)])
    tri2 = remap[tri].astype(np.uint32)
    cases = [(pts2, tri2, 3), (np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0.0]]), np.array([[0, 1, 2]], np.uint32), 1),
             (pts, tri, 1)]
    for p, t, nparts in cases:
        case = Case(p, t, nparts, seed=11)
        st = case.ostats(ol.F64)
        dt = 0.1 * st.maxEdgeLen ** 2
        marr = case.model.as_array()
        n_ref, c_ref = st.step(case.n0, case.c0, dt, 6, marr)
        assert np.isfinite(n_ref).all()
        case.mesh.init_stats(fa.PREC_F64)
        for ksteps in (0, 1, 4, 8):
            plan = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps)
            plan.set_state(case.n0, case.c0)
            plan.run(6, dt, case.model)
            n, c = plan.get_state()
            parity(n, c, n_ref, c_ref, TOL_F64, what=str((len(p), ksteps)))
        lay, dev = _ref_device(case)
        dev.set_state(case.n0, case.c0)
        dev.run(6, dt, case.model)
        n, c = dev.get_state()
        n_m, c_m = case.ostats(ol.MIXED).step(case.n0, case.c0, dt, 6, marr)
        parity(n, c, n_m, c_m, TOL_REF, what=str(len(p)))


# partition sizes bench.py pairs with each fused depth (rows per partition), by register tile
_TIMED_N0 = {8: {5: 1200, 6: 1050, 7: 975}, 2: {5: 880, 6: 760, 7: 650}}


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("ksteps", [5, 6, 7])
@pytest.mark.parametrize("cfg", [2, 8])
def test_timed_configuration_vs_oracle(cfg, ksteps, prec, monkeypatch):
    """the configuration bench.py times -- fused depth 5/6/7 on the 512 x 3 and 512 x 4 register tiles with
    the partition sizes the bench pairs with them, and the nontemporal prologue (FEM_TILE_PROLOGUE=3,
    otherwise only chosen when the operator exceeds the Infinity Cache) forced -- against the f64 oracle,
    on a regular sphere and on the irregular mesh, over a step count that is not a multiple of the
    fused depth (two full launches and a 3-step one)"""
    monkeypatch.setenv("FEM_TILE_CONFIG", str(cfg))
    monkeypatch.setenv("FEM_TILE_PROLOGUE", "3")
    from helpers import Case
    steps = 2 * ksteps + 3
    n0rows = _TIMED_N0[cfg][ksteps]
    sp, st_ = fa.make_icosphere(64, 0.1), fa.make_scanned_like(40, 0.1, seed=3, high_valence_fraction=0.01)
    for (pts, tri), regular in ((sp, True), (st_, False)):
        case = Case(pts, tri, max(2, len(pts) // n0rows))
        case.mesh.init_stats(fa.PREC_F64)
        st = case.ostats(ol.F64)
        dt = case.dt if regular else min(case.dt, stable_dt(st, case.model))
        plan = fa.Plan(case.mesh, fa.PREC_F64 if prec == "f64" else fa.PREC_F32, part=case.part, ksteps=ksteps)
        info = plan.tile_info()
        assert info["ksteps"] == ksteps and info["threads"] * info["rows_per_thread"] == {2: 1536, 8: 2048}[cfg]
        n0, c0 = (case.n0, case.c0) if regular else smooth_ic(case.pts)
        if prec == "f32":
            n0, c0 = n0.astype(np.float32).astype(np.float64), c0.astype(np.float32).astype(np.float64)
        plan.set_state(n0, c0)
        plan.run(steps, dt, case.model)
        n_gpu, c_gpu = plan.get_state()
        n_ref, c_ref = st.step(n0, c0, dt, steps, case.model.as_array())
        tol = TOL_F64 if prec == "f64" else TOL_F32
        parity(n_gpu, c_gpu, n_ref, c_ref, tol, what=str((regular, rel_l2(n_gpu, n_ref), rel_l2(c_gpu, c_ref))))
        assert not np.array_equal(n_gpu, n0)


@pytest.mark.parametrize("cfg,ksteps", [(8, 7), (2, 6)])
def test_packed_operator_records(cfg, ksteps, monkeypatch):
    """12-byte operator records (fp32 weight + three 21-bit gradient coefficients against a per-row scale), the default
    of fused fp32 plans without wide rows: the decoded coefficients are within 2^-21 of the row's largest gradient
    coefficient of the fp32 ones and the weights are bit-equal; the run agrees with the fp32-record plan to 1e-8 (a
    thousandth of the tolerance), is no further from the f64 oracle than that plan, and within the fp32 tolerance; plans that
    cannot use them (f64, device assembly) keep four scalars; meshes with wide rows keep four floats for the extras only"""
    from helpers import Case
    monkeypatch.setenv("FEM_TILE_CONFIG", str(cfg))
    monkeypatch.setenv("FEM_TILE_PROLOGUE", "3")
    pts, tri = fa.make_icosphere(64, 0.1)
    case = Case(pts, tri, max(2, len(pts) // _TIMED_N0[cfg][ksteps]))
    case.mesh.init_stats(fa.PREC_F64)
    packed = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=ksteps)
    monkeypatch.setenv("FEM_TILE_PACKED", "0")
    plain = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=ksteps)
    monkeypatch.delenv("FEM_TILE_PACKED")
    assert packed.operator_format() == 1 and plain.operator_format() == 0
    assert packed.operator_bytes() < 0.86 * plain.operator_bytes()
    a, b = packed.coef(), plain.coef()
    assert a.shape == b.shape and np.array_equal(a[:, 0], b[:, 0])
    big = np.abs(b[:, 1:]).max()
    assert np.abs(a[:, 1:] - b[:, 1:]).max() <= big * 2.0 ** -21 * 1.01 and np.abs(a[:, 1:] - b[:, 1:]).max() > 0
    steps = 14 * ksteps + 3
    n0, c0 = case.n0.astype(np.float32).astype(np.float64), case.c0.astype(np.float32).astype(np.float64)
    out = []
    for plan in (packed, plain):
        plan.set_state(n0, c0)
        plan.run(steps, case.dt, case.model)
        out.append(plan.get_state())
    n_ref, c_ref = case.ostats(ol.F64).step(n0, c0, case.dt, steps, case.model.as_array())
    e_packed, e_plain = rel_l2(out[0][0], n_ref), rel_l2(out[1][0], n_ref)
    parity(out[0][0], out[0][1], n_ref, c_ref, TOL_F32)
    # observed: packed vs fp32 records 2e-9, either against the oracle 2.4e-9 (1M vertices, 25..3200 steps: 1e-9..3e-9
    # between the two, 1e-8..2e-7 against the oracle for both, equal to four digits: tests/report_packed_accuracy.py)
    assert rel_l2(out[0][0], out[1][0]) <= 1e-8 and e_packed <= 2 * e_plain + 1e-9, (e_packed, e_plain, rel_l2(out[0][0], out[1][0]))
    # the single-step (sliced-ELL) plan reads the same records
    sell = fa.Plan(case.mesh, fa.PREC_F32)
    monkeypatch.setenv("FEM_TILE_PACKED", "0")
    sell_plain = fa.Plan(case.mesh, fa.PREC_F32)
    monkeypatch.delenv("FEM_TILE_PACKED")
    assert sell.operator_format() == 1 and sell_plain.operator_format() == 0 and sell.operator_bytes() < 0.86 * sell_plain.operator_bytes()
    a, b = sell.coef(), sell_plain.coef()
    assert a.shape == b.shape and np.array_equal(a[:, 0], b[:, 0]) and 0 < np.abs(a[:, 1:] - b[:, 1:]).max() <= big * 2.0 ** -21 * 1.01
    res = []
    for plan in (sell, sell_plain):
        plan.set_state(n0, c0)
        plan.run(steps, case.dt, case.model)
        res.append(plan.get_state())
    parity(res[0][0], res[0][1], n_ref, c_ref, TOL_F32)
    assert rel_l2(res[0][0], res[1][0]) <= 1e-8 and rel_l2(res[0][0], n_ref) <= 2 * rel_l2(res[1][0], n_ref) + 1e-9
    # plans that keep four scalars per entry
    assert fa.Plan(case.mesh, fa.PREC_F32, implicit=True).operator_format() == 0
    assert fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps).operator_format() == 0
    assert fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=ksteps, assemble=True).operator_format() == 0
    # rows wider than the register tile's six entries: their first six entries in records, the rest four floats
    wp, wt = fa.make_scanned_like(40, 0.1, seed=3, high_valence_fraction=0.01)
    wide = Case(wp, wt, max(2, len(wp) // 700))
    wide.mesh.init_stats(fa.PREC_F64)
    st = wide.ostats(ol.F64)
    dt = min(wide.dt, stable_dt(st, wide.model))
    n0, c0 = smooth_ic(wide.pts)
    n0, c0 = n0.astype(np.float32).astype(np.float64), c0.astype(np.float32).astype(np.float64)
    wplan = fa.Plan(wide.mesh, fa.PREC_F32, part=wide.part, ksteps=5)
    monkeypatch.setenv("FEM_TILE_PACKED", "0")
    wplain = fa.Plan(wide.mesh, fa.PREC_F32, part=wide.part, ksteps=5)
    monkeypatch.delenv("FEM_TILE_PACKED")
    assert wplan.operator_format() == 1 and wplain.operator_format() == 0
    a, b = wplan.coef(), wplain.coef()
    assert np.array_equal(a[:, 0], b[:, 0]) and np.abs(a[:, 1:] - b[:, 1:]).max() <= np.abs(b[:, 1:]).max() * 2.0 ** -21 * 1.01
    res = []
    for plan in (wplan, wplain):
        plan.set_state(n0, c0)
        plan.run(33, dt, wide.model)
        res.append(plan.get_state())
    n_ref, c_ref = st.step(n0, c0, dt, 33, wide.model.as_array())
    parity(res[0][0], res[0][1], n_ref, c_ref, TOL_F32)
    assert rel_l2(res[0][0], res[1][0]) <= 1e-7, rel_l2(res[0][0], res[1][0])


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_fused_plan_on_multilevel_partition(prec):
    """the METIS-style R0 stage (multilevel k-way on the dual graph: parts of unequal size, some needing a split to
    fit a tile) feeding the fused plan and the drop-in step, regular and irregular mesh, against the oracle"""
    from helpers import Case
    for (pts, tri), regular in ((fa.make_icosphere(56, 0.1), True), (fa.make_scanned_like(40, 0.1, seed=3, high_valence_fraction=0.01), False)):
        case = Case(pts, tri, max(2, len(pts) // 800), method="multilevel")
        sizes = np.diff(case.part.node_parts.astype(np.int64))
        assert sizes.min() > 0 and sizes.max() > sizes.min()
        case.mesh.init_stats(fa.PREC_F64)
        st = case.ostats(ol.F64)
        dt = case.dt if regular else min(case.dt, stable_dt(st, case.model))
        n0, c0 = smooth_ic(case.pts)
        if prec == "f32":
            n0, c0 = n0.astype(np.float32).astype(np.float64), c0.astype(np.float32).astype(np.float64)
        n_ref, c_ref = st.step(n0, c0, dt, 13, case.model.as_array())
        plan = fa.Plan(case.mesh, fa.PREC_F64 if prec == "f64" else fa.PREC_F32, part=case.part, ksteps=5)
        plan.set_state(n0, c0)
        plan.run(13, dt, case.model)
        n_gpu, c_gpu = plan.get_state()
        tol = TOL_F64 if prec == "f64" else TOL_F32
        parity(n_gpu, c_gpu, n_ref, c_ref, tol, tol_dev=None if prec == "f64" else 1e-4)   # smooth IC, 13 steps: observed 1.1e-5
    if prec == "f64":
        case = Case(*fa.make_icosphere(24, 0.1), 24, method="multilevel")
        lay, dev = _ref_device(case)
        dev.set_state(case.n0, case.c0)
        dev.run(4, case.dt, case.model)
        n_gpu, c_gpu = dev.get_state()
        n_ref, c_ref = case.ostats(ol.MIXED).step(case.n0, case.c0, case.dt, 4, case.model.as_array())
        parity(n_gpu, c_gpu, n_ref, c_ref, TOL_REF)


def test_scan_pathologies_all_paths(tmp_path):
    """BASELINE config 4 with what REAL scans bring (helpers.pathological_scan: open boundaries, vertices without
    faces, a non-manifold bow-tie vertex, valences 3..16), read back from a scanner-style binary PLY (normals and
    colours to skip) and upsampled once: the drop-in `step`, the sliced-ELL plan and the fused plan (f64 and f32, wide
    rows through the LDS extras, boundary rows with open fans) against the oracle.  No real scan data can be fetched
    here, so parity on actual Lucy/bunny files stays unpinned; this pins the code paths they would take."""
    from helpers import Case, pathological_scan, write_binary_ply
    pts, tri = pathological_scan(28)
    path = str(tmp_path / "scan.ply")
    write_binary_ply(path, pts, tri)
    p2, t2, _ = fa.read_mesh(path)
    assert np.array_equal(t2, tri) and np.array_equal(p2, pts.astype(np.float32).astype(np.float64))
    p2, t2 = fa.subdivide(p2, t2, 1)
    case = Case(p2, t2, max(2, len(p2) // 400))            # <= 1024 faces per partition: the reference layout's limit
    assert case.mesh.H - 3 * case.mesh.F > 100             # boundary half-edges
    deg = np.diff(case.topo.rowptr)
    assert (deg == 0).sum() > 10 and deg.max() >= 12       # vertices without faces; wide rows
    st = case.ostats(ol.F64)
    dt = stable_dt(st, case.model)
    n0, c0 = smooth_ic(case.pts)
    marr = case.model.as_array()
    case.mesh.init_stats(fa.PREC_MIXED)
    lay, dev = _ref_device(case)
    dev.set_state(n0, c0)
    dev.run(6, dt, case.model)
    n_gpu, c_gpu = dev.get_state()
    n_ref, c_ref = case.ostats(ol.MIXED).step(n0, c0, dt, 6, marr)
    parity(n_gpu, c_gpu, n_ref, c_ref, TOL_REF)
    case.mesh.init_stats(fa.PREC_F64)
    n_ref, c_ref = st.step(n0, c0, dt, 23, marr)
    for ksteps in (0, 5):
        plan = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps)
        plan.set_state(n0, c0)
        plan.run(23, dt, case.model)
        n_gpu, c_gpu = plan.get_state()
        parity(n_gpu, c_gpu, n_ref, c_ref, TOL_F64, what="f64 k=%d" % ksteps)
    n0r, c0r = n0.astype(np.float32).astype(np.float64), c0.astype(np.float32).astype(np.float64)
    n_ref, c_ref = st.step(n0r, c0r, dt, 23, marr)
    for ksteps in (0, 5):
        plan = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=ksteps)
        plan.set_state(n0r, c0r)
        plan.run(23, dt, case.model)
        n_gpu, c_gpu = plan.get_state()
        parity(n_gpu, c_gpu, n_ref, c_ref, TOL_F32, tol_dev=DEV_SMOOTH_LARGE, what="f32 k=%d" % ksteps)


@pytest.mark.gpu
def test_wide_rows_first_numbering_on_the_fused_plan(monkeypatch):
    """the drivers' numbering pre-pass for irregular meshes (fem_mesh_wide_rows_first: rows of more than 6 entries lead
    every part, ring rows ordered the same way inside the tiles) changes which rows share a wave, not what is computed:
    the fused fp32 and fp64 plans built through it agree with the f64 oracle on the re-ordered mesh, and -- mapped back
    to the caller's numbering through perm_v -- with the run that keeps the input numbering (same smooth initial
    condition, defined on the caller's vertices).  Observed on the deviation: f32 2e-7 against the oracle, 1e-7 between
    the two numberings (the entries of a row are summed in another order); f64 1e-13."""
    from fem_amd import dist as fdist
    pts, tri = fa.make_scanned_like(40, 0.1, seed=3, high_valence_fraction=0.01)
    n_in, c_in = smooth_ic(pts)
    steps, back = 13, {}
    for prec, tol in (("f32", TOL_F32), ("f64", TOL_F64)):
        for wide_first in ("1", "0"):
            monkeypatch.setenv("FEM_WIDE_FIRST", wide_first)
            run = fdist.SingleRun(fa.Mesh(pts, tri), len(pts) // 820, prec, ksteps=5, dt_rule="stable")
            om, pv = run.part.ordered, run.part.perm_v
            assert np.array_equal(om.points[pv], pts)
            ex = np.maximum(np.diff(om.topology()["rowptr"].astype(np.int64)) - 6, 0)
            lead = all(np.all(np.diff(ex[run.part.node_parts[p]:run.part.node_parts[p + 1]]) <= 0) for p in range(run.part.n))
            assert lead == (wide_first == "1") and ex.max() >= 4
            n0, c0 = np.empty(om.V), np.empty(om.V)
            n0[pv], c0[pv] = n_in, c_in
            if prec == "f32":
                n0, c0 = n0.astype(np.float32).astype(np.float64), c0.astype(np.float32).astype(np.float64)
            run.plan.set_state(n0, c0)
            run.run(steps)
            n_gpu, c_gpu = run.plan.get_state()
            st = ol.Stats(ol.Topology(om.V, om.tri), om.points, ol.F64)
            n_ref, c_ref = st.step(n0, c0, run.dt, steps, run.model.as_array())
            parity(n_gpu, c_gpu, n_ref, c_ref, tol, what="%s wide_first=%s" % (prec, wide_first))
            assert not np.array_equal(n_gpu, n0)
            back[(prec, wide_first)] = (n_gpu[pv], c_gpu[pv])
        (na, ca), (nb, cb) = back[(prec, "1")], back[(prec, "0")]
        parity(na, ca, nb, cb, tol, what="%s: the two numberings" % prec)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,ksteps", [(8, 7), (2, 5)])
def test_mixed_records_plan(cfg, ksteps, monkeypatch):
    """FEM_PLAN_MIXED_RECORDS: a fused fp64 plan in the reference's own precision mix (fp64 state, differences, Laplacian
    and update; gradient coefficients, narrowed differences and gradient sums in fp32: FEM_common.cu:99-158).  The
    downloaded operator is the fp64 plan's with the three gradient coefficients of the register rows rounded to fp32
    (weights and the extras of wide rows bit-equal); the run stays within 1e-9 / 1e-7 (fields / deviation) of the f64
    oracle and of the oracle's MIXED CPUrun, two orders closer than the fp32 plan; other plan kinds refuse the flag.
    Observed against the f64 oracle: fields 9e-11, deviation 1.7e-9 on the sphere (random IC), 4e-15 / 2e-12 on the
    irregular mesh (smooth IC); fp32 plan: 4e-9 / 1e-7, fp64 plan: 1e-14 / 1e-12."""
    monkeypatch.setenv("FEM_TILE_CONFIG", str(cfg))
    from helpers import Case
    steps = 2 * ksteps + 3
    for (pts, tri), regular in ((fa.make_icosphere(64, 0.1), True),
                                (fa.make_scanned_like(40, 0.1, seed=3, high_valence_fraction=0.01), False)):
        case = Case(pts, tri, max(2, len(pts) // _TIMED_N0[cfg][ksteps]))
        case.mesh.init_stats(fa.PREC_F64)
        st = case.ostats(ol.F64)
        dt = case.dt if regular else min(case.dt, stable_dt(st, case.model))
        full = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps)
        mixed = fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=ksteps, mixed=True)
        assert mixed.operator_format() == 2 and full.operator_format() == 0
        assert mixed.operator_bytes() < 0.75 * full.operator_bytes()
        cf, cm = full.coef().reshape(-1, 4), mixed.coef().reshape(-1, 4)
        assert np.array_equal(cm[:, 0], cf[:, 0])
        rounded = cf[:, 1:].astype(np.float32).astype(np.float64)
        same = np.all(cm[:, 1:] == rounded, axis=1) | np.all(cm[:, 1:] == cf[:, 1:], axis=1)     # register rows | extras
        assert same.all() and (np.all(cm[:, 1:] == cf[:, 1:], axis=1).mean() < 0.9)
        n0, c0 = (case.n0, case.c0) if regular else smooth_ic(case.pts)
        mixed.set_state(n0, c0)
        mixed.run(steps, dt, case.model)
        n_gpu, c_gpu = mixed.get_state()
        n_ref, c_ref = st.step(n0, c0, dt, steps, case.model.as_array())
        e = parity(n_gpu, c_gpu, n_ref, c_ref, 1e-9, tol_dev=1e-7, what="mixed vs f64 oracle")
        n_mx, c_mx = case.ostats(ol.MIXED).step(n0, c0, dt, steps, case.model.as_array())
        # (the MIXED oracle also assembles from fp32 geometry, mesh_function.cpp:119-121: an operator 1e-7 away)
        e2 = parity(n_gpu, c_gpu, n_mx, c_mx, 1e-7, tol_dev=1e-5 if regular else 2e-4, what="mixed vs MIXED oracle")
        assert not np.array_equal(n_gpu, n0)
        print("mixed records, regular=%s: vs f64 oracle %s, vs MIXED oracle %s" % (regular, e, e2))
    for kw in (dict(precision=fa.PREC_F32, ksteps=ksteps), dict(precision=fa.PREC_F64, ksteps=0)):
        with pytest.raises(fa.FemError):
            fa.Plan(case.mesh, kw["precision"], part=case.part, ksteps=kw["ksteps"], mixed=True)
