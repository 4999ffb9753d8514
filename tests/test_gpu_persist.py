"""tile_step_persist_kernel (fem_amd/csrc/kernels_step.hpp): several groups of k fused steps per launch, the tiles handed
from group to group through per-tile flags inside the launch instead of through a launch boundary.  The kernel runs the
same tile body in the same order per row as tile_step_kernel, so its state must equal the launch-per-group state BIT
FOR BIT (one stale neighbour row, read before its owner published it, would show), and it is checked against the f64
oracle like every other path.  No counterpart in the reference (one launch per step, src/main.cpp:558-562)."""
import numpy as np
import pytest

import fem_amd as fa
import oracle_lib as ol
from helpers import Case, parity, smooth_ic, stable_dt

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-5


def _plans(case, ksteps, monkeypatch, groups, cfg=None):
    """(launch-per-group plan, persistent plan) of the same mesh"""
    if cfg is not None:
        monkeypatch.setenv("FEM_TILE_CONFIG", str(cfg))
    monkeypatch.setenv("FEM_TILE_PERSIST", "0")
    plain = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=ksteps)
    monkeypatch.setenv("FEM_TILE_PERSIST", str(groups))
    pers = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=ksteps)
    monkeypatch.delenv("FEM_TILE_PERSIST")
    assert plain.persist_groups() == 0 and pers.persist_groups() == groups
    assert pers.tile_info()["groups_per_launch"] == groups and plain.tile_info()["groups_per_launch"] == 1
    return plain, pers


@pytest.mark.parametrize("cfg,ksteps,n0rows,groups", [(8, 7, 975, 3), (2, 6, 760, 2), (8, 5, 400, 9)])
def test_persistent_launches_equal_launch_per_group_and_oracle(cfg, ksteps, n0rows, groups, monkeypatch):
    """regular sphere and irregular mesh (rows wider than the register tile: the X = 2 build), both default register
    tiles, over a step count that takes several persistent launches, a last group shorter than k and a last launch of a
    single group (which goes to the plain kernel): bit-identical to one launch per group, and within the fp32 bars of
    the f64 oracle"""
    sp, sc = fa.make_icosphere(64, 0.1), fa.make_scanned_like(40, 0.1, seed=3, high_valence_fraction=0.01)
    steps = (2 * groups + 1) * ksteps + 3
    for (pts, tri), regular in ((sp, True), (sc, False)):
        case = Case(pts, tri, max(2, len(pts) // n0rows))
        case.mesh.init_stats(fa.PREC_F64)
        st = case.ostats(ol.F64)
        dt = case.dt if regular else min(case.dt, stable_dt(st, case.model))
        n0, c0 = case.n0.astype(np.float32).astype(np.float64), case.c0.astype(np.float32).astype(np.float64)
        plain, pers = _plans(case, ksteps, monkeypatch, groups, cfg)
        assert pers.tile_info()["ntiles"] >= 8
        out = []
        for plan in (plain, pers):
            plan.set_state(n0, c0)
            plan.run(steps, dt, case.model)
            out.append(plan.get_state())
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), (
            "rows differing: %d" % int((out[0][0] != out[1][0]).sum()))
        assert not np.array_equal(out[1][0], n0)
        n_ref, c_ref = st.step(n0, c0, dt, steps, case.model.as_array())
        parity(out[1][0], out[1][1], n_ref, c_ref, TOL_F32, what=str((cfg, regular)))
        # a second run on the same plan (flags and tickets start from zero in every launch), in pieces
        pers.set_state(n0, c0)
        pers.run(2 * ksteps, dt, case.model)
        pers.run(steps - 2 * ksteps, dt, case.model)
        again = pers.get_state()
        assert np.array_equal(again[0], out[0][0]) and np.array_equal(again[1], out[0][1])


@pytest.mark.parametrize("nu,nparts", [(8, 1), (16, 2), (20, 5), (24, 7)])
def test_persistent_launches_with_fewer_tiles_than_ranges(nu, nparts, monkeypatch):
    """1, 2, 5 and 7 tiles: some of the kernel's eight tile ranges are empty (their workgroups leave at once) and the
    grid is the minimum of eight workgroups; one tile depends on nothing but itself"""
    case = Case(*fa.make_icosphere(nu, 0.1), nparts)
    case.mesh.init_stats(fa.PREC_F64)
    plain, pers = _plans(case, 4, monkeypatch, 5, 8)
    assert pers.tile_info()["ntiles"] == nparts
    n0, c0 = case.n0.astype(np.float32).astype(np.float64), case.c0.astype(np.float32).astype(np.float64)
    out = []
    for plan in (plain, pers):
        plan.set_state(n0, c0)
        plan.run(43, case.dt, case.model)
        out.append(plan.get_state())
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    n_ref, c_ref = case.ostats(ol.F64).step(n0, c0, case.dt, 43, case.model.as_array())
    parity(out[1][0], out[1][1], n_ref, c_ref, TOL_F32)


def test_persistent_rule_by_tile_count_and_a_mid_sized_run(monkeypatch):
    """the library's own choice (fem_plan_persist_groups): plans of 192..2400 tiles run 12..32 groups per launch,
    smaller and larger ones and every plan that is not fp32 with packed records one launch per group; and the size the
    rule exists for -- 256k rows, 263 tiles on 512 workgroup slots, 101 steps in one persistent launch -- against the
    oracle and against launch-per-group"""
    monkeypatch.delenv("FEM_TILE_PERSIST", raising=False)
    monkeypatch.setenv("FEM_TILE_CONFIG", "8")
    small = Case(*fa.make_icosphere(32, 0.1), 10)
    small.mesh.init_stats(fa.PREC_F64)
    assert fa.Plan(small.mesh, fa.PREC_F32, part=small.part, ksteps=7).persist_groups() == 0
    pts, tri = fa.make_icosphere(160, 0.1)
    case = Case(pts, tri, len(pts) // 975)
    case.mesh.init_stats(fa.PREC_F64)
    plan = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=7)
    assert 192 <= plan.tile_info()["ntiles"] <= 2400 and plan.persist_groups() == 32
    assert fa.Plan(case.mesh, fa.PREC_F64, part=case.part, ksteps=7).persist_groups() == 0
    monkeypatch.setenv("FEM_TILE_PACKED", "0")
    assert fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=7).persist_groups() == 0
    monkeypatch.delenv("FEM_TILE_PACKED")
    monkeypatch.setenv("FEM_TILE_PERSIST", "0")
    plain = fa.Plan(case.mesh, fa.PREC_F32, part=case.part, ksteps=7)
    assert plain.persist_groups() == 0
    n0, c0 = case.n0.astype(np.float32).astype(np.float64), case.c0.astype(np.float32).astype(np.float64)
    steps = 101
    out = []
    for p in (plain, plan):
        p.set_state(n0, c0)
        p.run(steps, case.dt, case.model)
        out.append(p.get_state())
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    n_ref, c_ref = case.ostats(ol.F64).step(n0, c0, case.dt, steps, case.model.as_array())
    parity(out[1][0], out[1][1], n_ref, c_ref, TOL_F32)


def test_persistent_launches_on_a_shard_plan(monkeypatch):
    """a shard's plan (tiles [interior | boundary | ghost], ghost rows in tiles of their own, rings cut off at the
    shard's edge) through fem_plan_run: 2 x 4 fused steps inside 8 ghost rings in ONE persistent launch, exact on the
    owned rows against the launch-per-group plan"""
    case = Case(*fa.make_icosphere(48, 0.1), 60)
    case.mesh.init_stats(fa.PREC_F64)
    shard = fa.Shard(case.part, 1, 3, 8)
    l2g = shard.loc2glob
    monkeypatch.setenv("FEM_TILE_CONFIG", "8")
    out = []
    for groups in ("0", "2"):
        monkeypatch.setenv("FEM_TILE_PERSIST", groups)
        sp = fa.Plan(case.mesh, fa.PREC_F32, shard=shard, part=case.part, ksteps=4)
        assert sp.persist_groups() == int(groups)
        sp.set_state(case.n0[l2g], case.c0[l2g])
        sp.run(8, case.dt, case.model)
        out.append(sp.get_state())
    own = slice(0, shard.n_own)
    assert np.array_equal(out[0][0][own], out[1][0][own]) and np.array_equal(out[0][1][own], out[1][1][own])
    n_ref, c_ref = case.ostats(ol.F64).step(case.n0, case.c0, case.dt, 8, case.model.as_array())
    parity(out[1][0][own], out[1][1][own], n_ref[l2g[own]], c_ref[l2g[own]], TOL_F32)
