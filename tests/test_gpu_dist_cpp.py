"""The C/C++ multi-GPU host (fem_amd/csrc/dist.cpp = libfem_dist.so, driven by `fem_run --ranks N`): the halo
exchange over RCCL and the schedule that overlaps it with the interior tiles, with no Python in the loop.
On the one GPU a test box has: RCCL itself at world size 1 through every branch of the schedule, and the
multi-rank exchange plan with the ranks as threads sharing the GPU (device-to-device copies between the
ranks' buffers); both against the f64 oracle, and against the Python driver (fem_amd/dist.py:ShardedRun)."""
import os
import subprocess

import numpy as np
import pytest

import fem_amd as fa
from helpers import parity, rel_l2  # noqa: F401
from test_sharded import launch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FEM_RUN = os.path.join(ROOT, "fem_amd", "lib", "fem_run")
NU, NPARTS, ITER = 24, 40, 9              # V = 5762; 2 + 2 * 9 = 20 steps


@pytest.fixture(scope="module")
def sphere_off(tmp_path_factory):
    pts, tri = fa.make_icosphere(NU, 0.1)
    path = str(tmp_path_factory.mktemp("dist_cpp") / "sphere.off")
    fa.off_write(path, pts, tri)
    return path


@pytest.fixture(scope="module")
def reference(sphere_off):
    """oracle run on the mesh as the file holds it (the .off text rounds the coordinates)"""
    import oracle_lib as ol
    pts, tri, _ = fa.off_read(sphere_off)
    part = fa.Mesh(pts, tri).partition(NPARTS)
    om = part.ordered
    st = ol.Stats(ol.Topology(om.V, om.tri), om.points, ol.F64)
    n0, c0 = fa.initial_condition(om.V, 3)
    return part, st.step(n0, c0, 0.1 * st.maxEdgeLen ** 2, 2 + 2 * ITER, fa.default_model().as_array())


def run_fem(cmd, env=None):
    """fem_run as a child process.  A run that does not finish says where it stopped instead of dying silently at the
    time limit: --watchdog makes the program report every rank's stage and exit 124, FEM_DIST_TRACE adds one stderr
    line per exchange stage, and the barrier of the in-process world gives up after 40 s."""
    env = dict(os.environ if env is None else env)
    env.update(FEM_DIST_TRACE="1", FEM_DIST_BARRIER_SECONDS="40")
    try:
        res = subprocess.run(cmd + ["--watchdog", "90"], capture_output=True, text=True, timeout=150, env=env)
    except subprocess.TimeoutExpired as e:
        pytest.fail("fem_run did not finish: %s\n%s" % (" ".join(cmd), ((e.stderr or b"")[-3000:]).decode(errors="replace") if isinstance(e.stderr, bytes) else str(e.stderr)[-3000:]))
    assert res.returncode == 0 and "Best time" in res.stdout, "exit %d\n%s\n%s" % (res.returncode, res.stdout[-1500:], res.stderr[-3000:])
    return res


def run_ranks(path, tmp_path, ranks, fuse, rings, transport, overlap=None, f64=True, mixed=False):
    dump = str(tmp_path / "state.bin")
    cmd = [FEM_RUN, path, "--ranks", str(ranks), "-n", str(NPARTS), "--fuse", str(fuse), "--rings", str(rings),
           "--iterations", str(ITER), "--seed", "3", "--transport", transport, "--dump", dump] + (
               ["--mixed"] if mixed else (["--f64"] if f64 else []))
    env = dict(os.environ)
    env.pop("FEM_OVERLAP", None)
    if overlap is not None:
        env["FEM_OVERLAP"] = overlap
    res = run_fem(cmd, env)
    raw = np.fromfile(dump)
    V = len(raw) // 2
    return raw[:V], raw[V:], res.stdout


@pytest.mark.parametrize("fuse,rings,overlap", [(4, 8, None), (4, 4, "split"), (4, 8, "sync"), (4, 6, "0"), (0, 5, None)])
def test_rccl_world_of_one_cpp(sphere_off, reference, tmp_path, fuse, rings, overlap):
    """ncclCommInitRank / ncclGroupStart / ncclGroupEnd and the event-ordered schedule at world size 1: merged and
    interior-first groups, the blocking mode, the plain schedule, the sliced-ELL shard plan"""
    part, (n_ref, c_ref) = reference
    n, c, out = run_ranks(sphere_off, tmp_path, 1, fuse, rings, "rccl", overlap)
    assert "RCCL" in out
    parity(n, c, n_ref, c_ref, 1e-11)


@pytest.mark.parametrize("ranks,fuse,rings,overlap", [(2, 4, 8, None), (3, 4, 4, "split"), (3, 3, 7, None), (4, 0, 3, None),
                                                       (3, 4, 8, "0")])
def test_ranks_as_threads_on_one_gpu(sphere_off, reference, tmp_path, ranks, fuse, rings, overlap):
    """the multi-rank exchange plan (send blocks by destination, ghost segments by owner, k rings = k steps between
    exchanges, boundary tiles before the pack, ghost tiles skipped) with real neighbours: stale or missing ghost
    rows would show in the owned rows within a few steps"""
    part, (n_ref, c_ref) = reference
    n, c, out = run_ranks(sphere_off, tmp_path, ranks, fuse, rings, "inproc", overlap)
    assert "Ranks: %d (in-process copies" % ranks in out
    parity(n, c, n_ref, c_ref, 1e-11)


@pytest.mark.parametrize("ranks,fuse,rings,packed", [(2, 4, 8, None), (3, 4, 4, None), (3, 4, 4, "0")])
def test_fp32_shards_against_the_f64_oracle(sphere_off, reference, tmp_path, ranks, fuse, rings, packed, monkeypatch):
    """fp32 shard plans -- 12-byte operator records in boundary, ghost and interior tiles by default, four floats with
    FEM_TILE_PACKED=0 -- against the f64 oracle (not against another packed plan, where a packing fault in the ghost or
    boundary tiles would cancel out): the fp32 bar on the fields, and on the deviation (random IC, 20 steps; observed
    n 1.5e-7, c 9e-8 of the deviation with either record format)"""
    part, (n_ref, c_ref) = reference
    if packed is not None:
        monkeypatch.setenv("FEM_TILE_PACKED", packed)
    n, c, out = run_ranks(sphere_off, tmp_path, ranks, fuse, rings, "inproc", None, f64=False)
    assert "Ranks: %d (in-process copies" % ranks in out
    parity(n, c, n_ref, c_ref, 1e-5)
    assert rel_l2(n, n_ref) > 1e-12                 # fp32 did run (an f64 plan would be at 1e-14)


def test_mixed_record_shards_against_the_f64_oracle(sphere_off, reference, tmp_path):
    """fp64 shard plans in the reference's own precision mix (fem_run --mixed: FEM_PLAN_MIXED_RECORDS in boundary, ghost and
    interior tiles), three ranks with real neighbours: between the fp64 plans (1e-14) and the fp32 ones (1e-7 of the
    deviation) -- observed n 2e-9 of the deviation"""
    part, (n_ref, c_ref) = reference
    n, c, out = run_ranks(sphere_off, tmp_path, 3, 4, 8, "inproc", None, mixed=True)
    assert "Ranks: 3 (in-process copies" in out
    parity(n, c, n_ref, c_ref, 1e-9, tol_dev=1e-7)
    assert rel_l2(n, n_ref) > 1e-13                 # the fp32 gradient branch did run (an f64 plan would be at 1e-14)


def test_rccl_refuses_more_ranks_than_gpus(sphere_off):
    """two ranks of one RCCL communicator on the same GPU are refused up front (RCCL rejects or hangs on duplicates)"""
    cmd = [FEM_RUN, sphere_off, "--ranks", "2", "-n", str(NPARTS), "--fuse", "4", "--iterations", "2", "--transport", "rccl"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "needs that many GPUs" in res.stderr, res.stderr[-1500:]


def test_python_driver_agrees_with_cpp_host(sphere_off, reference, tmp_path):
    """fem_amd/dist.py:ShardedRun (torch.distributed, backend nccl, world 1) and fem_run --ranks 1 (RCCL from C++)
    drive the same plans through the same schedule: same state, fp32"""
    n_cpp, c_cpp, _ = run_ranks(sphere_off, tmp_path, 1, 4, 8, "rccl", None, f64=False)
    res = launch(1, tmp_path, backend="hip", sphere=NU, nparts=NPARTS, krings=8, steps=2 + 2 * ITER, precision="f32",
                 fuse=4, pg="nccl")[0]
    # (the Python worker builds the sphere in memory, fem_run reads its rounded .off text: 1e-9 apart, not bit-equal)
    parity(res["n"], res["c"], n_cpp, c_cpp, 1e-7)


@pytest.mark.parametrize("ranks,transport", [(1, "rccl"), (2, "inproc"), (4, "inproc")])
def test_sharded_semi_implicit_steps(sphere_off, tmp_path, ranks, transport):
    """semi-implicit stepping on shards (fem_dist_run_imex): conjugate gradients distributed over the ranks -- one ghost
    ring, an exchange of the right-hand side and of every search direction, the dot products over owned rows summed by
    an all-reduce (ncclAllReduce at world size 1 here; through the host between the in-process ranks) -- against the
    single-GPU solver (fem_plan_run_imex, itself checked against a scipy direct solve in test_gpu_implicit.py)"""
    iters, scale, cg = 3, 6.0, 30
    dump = str(tmp_path / "imex.bin")
    cmd = [FEM_RUN, sphere_off, "--ranks", str(ranks), "-n", str(NPARTS), "--iterations", str(iters), "--seed", "3", "--transport", transport,
           "--imex", str(cg), "--dt-scale", str(scale), "--f64", "--dump", dump]
    res = run_fem(cmd)
    raw = np.fromfile(dump)
    V = len(raw) // 2
    pts, tri, _ = fa.off_read(sphere_off)
    part = fa.Mesh(pts, tri).partition(NPARTS)
    om = part.ordered
    om.init_stats(fa.PREC_F64)
    dt = scale * 0.1 * om.stats()["maxEdgeLen"] ** 2
    plan = fa.Plan(om, fa.PREC_F64, implicit=True)
    n0, c0 = fa.initial_condition(om.V, 3)
    plan.set_state(n0, c0)
    plan.run_imex(2 + 2 * iters, dt, fa.default_model(), cg)
    n_ref, c_ref = plan.get_state()
    assert V == om.V and not np.array_equal(n_ref, n0)
    parity(raw[:V], raw[V:], n_ref, c_ref, 1e-10)


@pytest.mark.parametrize("ranks,transport,fuse", [(1, "rccl", 4), (2, "inproc", 4), (3, "inproc", 3)])
def test_sharded_semi_implicit_steps_on_fused_tiles(sphere_off, tmp_path, ranks, transport, fuse):
    """fem_dist_run_imex_fused: the Chebyshev solver on the shards' fused tiles -- every rank iterates its owned AND ghost
    rows, `fuse` iterations per launch, and fetches iterate and direction of its ghost rows after every launch; no
    all-reduce, the coefficient sequence follows from the whole mesh's largest row sum, which every rank is given --
    against the single-GPU run of the same solver (fem_plan_run_imex_fused, itself checked against scipy direct solves
    in test_gpu_implicit.py): the same iterations on the same numbers, so they agree to rounding"""
    iters, scale, tol = 3, 6.0, 1e-10
    dump = str(tmp_path / "imexf.bin")
    cmd = [FEM_RUN, sphere_off, "--ranks", str(ranks), "-n", str(NPARTS), "--fuse", str(fuse), "--iterations", str(iters), "--seed", "3",
           "--transport", transport, "--imex-fused", str(tol), "--dt-scale", str(scale), "--f64", "--dump", dump]
    run_fem(cmd)
    raw = np.fromfile(dump)
    V = len(raw) // 2
    pts, tri, _ = fa.off_read(sphere_off)
    part = fa.Mesh(pts, tri).partition(NPARTS)
    om = part.ordered
    om.init_stats(fa.PREC_F64)
    dt = scale * 0.1 * om.stats()["maxEdgeLen"] ** 2
    plan = fa.Plan(om, fa.PREC_F64, part=part, ksteps=fuse)
    n0, c0 = fa.initial_condition(om.V, 3)
    plan.set_state(n0, c0)
    plan.run_imex_fused(2 + 2 * iters, dt, fa.default_model(), rel_tol=tol)
    n_ref, c_ref = plan.get_state()
    assert V == om.V and not np.array_equal(n_ref, n0)
    parity(raw[:V], raw[V:], n_ref, c_ref, 1e-11)


def test_failing_rank_does_not_hang_the_others(sphere_off, tmp_path):
    """a rank whose exchange fails (injected: FEM_DIST_TEST_FAIL_RANK) breaks the in-process barrier: the other ranks
    return an error at once instead of waiting for it, and the program exits non-zero with both messages"""
    import time
    cmd = [FEM_RUN, sphere_off, "--ranks", "3", "-n", str(NPARTS), "--fuse", "4", "--rings", "4", "--iterations", str(ITER), "--seed", "3",
           "--transport", "inproc", "--f64", "--watchdog", "60"]
    env = dict(os.environ, FEM_DIST_TEST_FAIL_RANK="1", FEM_DIST_BARRIER_SECONDS="30")
    t0 = time.time()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode not in (0, 124) and time.time() - t0 < 25, (res.returncode, res.stderr[-2000:])
    assert "injected failure of rank 1" in res.stderr and "barrier broken" in res.stderr, res.stderr[-2000:]


def test_fused_tile_solver_refuses_fewer_rings_than_iterations_per_launch(sphere_off):
    """--imex-fused with --fuse 4 on shards of 2 ghost rings: a launch would iterate ghost rows that went stale two
    iterations earlier; the host refuses instead of computing on them"""
    cmd = [FEM_RUN, sphere_off, "--ranks", "2", "-n", str(NPARTS), "--fuse", "4", "--rings", "2", "--iterations", "2", "--transport", "inproc",
           "--imex-fused", "1e-8", "--f64", "--watchdog", "60"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "ghost rings" in res.stderr, (res.returncode, res.stderr[-1500:])
