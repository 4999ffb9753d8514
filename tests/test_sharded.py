"""Multi-GPU path: partition-range shards, k ghost rings, halo exchange.
CPU (gloo, world_size 2 and 3): the exchange plan + torch.distributed call pattern with the oracle as
the stepper (NaN-poisoned, so any missing/stale ghost is caught).  GPU: the HIP stepper, several
shards on one card -- in one process with a loop-back exchange, and as 2 gloo ranks."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

import fem_amd as fa
import oracle_lib as ol
from helpers import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_run(nu, nparts, steps, seed=3):
    pts, tri = fa.make_icosphere(nu, 0.1)
    part = fa.Mesh(pts, tri).partition(nparts)
    om = part.ordered
    st = ol.Stats(ol.Topology(om.V, om.tri), om.points, ol.F64)
    dt = 0.1 * st.maxEdgeLen ** 2
    n0, c0 = fa.initial_condition(om.V, seed)
    return part, st.step(n0, c0, dt, steps, fa.default_model().as_array())


_launches = [0]


def launch(world, tmp_path, **kw):
    out = str(tmp_path / "shard")
    _launches[0] += 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + (os.getpid() * 16 + _launches[0]) % 2000),
           os.path.join(ROOT, "tests", "dist_worker.py"), "--out", out]
    for k, v in kw.items():
        cmd += ["--" + k, str(v)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run(cmd, check=True, timeout=600, env=env, capture_output=True)
    files = sorted(glob.glob(out + ".*.npz"))
    assert len(files) == world
    return [np.load(f) for f in files]


@pytest.mark.parametrize("world,krings,steps", [(2, 1, 5), (2, 3, 7), (3, 2, 9)])
def test_gloo_oracle_stepper(world, krings, steps, tmp_path):
    nu, nparts = 8, 6
    part, (n_ref, c_ref) = reference_run(nu, nparts, steps)
    res = launch(world, tmp_path, backend="oracle", sphere=nu, nparts=nparts, krings=krings, steps=steps)
    covered = 0
    for r in res:
        b, e = int(r["own_begin"]), int(r["own_end"])
        assert not np.isnan(r["n"]).any() and not np.isnan(r["c"]).any()
        parity(r["n"], r["c"], n_ref[b:e], c_ref[b:e], 1e-13)
        covered += e - b
    assert covered == part.ordered.V


@pytest.mark.parametrize("world,krings,fuse,steps,overlap", [
    (2, 4, 2, 11, "sync"),     # blocking exchange: it has always arrived, merged first group
    (2, 4, 2, 11, "split"),    # interior first, wait, boundary + ghost
    (3, 3, 3, 10, "split"),    # one group per exchange
    (3, 3, 3, 10, "sync"),
    (2, 6, 4, 13, "1"),        # adaptive (gloo's asynchronous handle: either path, run to run); groups of 4 + 2
    (2, 4, 2, 11, "0"),        # plain schedule through the same stepper
])
def test_gloo_overlapped_schedule_on_oracle_tiles(world, krings, fuse, steps, overlap, tmp_path, monkeypatch):
    """the [interior | boundary | ghost] schedule of ShardedRun (fem_amd/dist.py) without a GPU:
    an oracle-backed stepper with two buffers, stale rows where a tile was not launched and NaN
    where nothing was ever written"""
    nu, nparts = 24, 10
    part, (n_ref, c_ref) = reference_run(nu, nparts, steps)
    monkeypatch.setenv("FEM_OVERLAP", overlap)
    res = launch(world, tmp_path, backend="oracle", sphere=nu, nparts=nparts, krings=krings, steps=steps, fuse=fuse)
    for r in res:
        b, e = int(r["own_begin"]), int(r["own_end"])
        assert not np.isnan(r["n"]).any() and not np.isnan(r["c"]).any()
        parity(r["n"], r["c"], n_ref[b:e], c_ref[b:e], 1e-13)
        if overlap == "0":
            assert tuple(r["classes"]) == (-1, -1, -1)
        else:
            nI, nB, nG = r["classes"]
            assert nI > 0 and nB > 0 and nG > 0
            ranges = {(int(f), int(c)) for f, c, _ in r["launches"]}
            assert (nI, nB) in ranges and (0, nI) in ranges        # boundary-only and interior-only launches happened
            if overlap == "split":
                assert (nI, nB + nG) in ranges
            elif overlap == "sync":
                assert (0, nI + nB + nG) in ranges
            else:
                assert (nI, nB + nG) in ranges or (0, nI + nB + nG) in ranges


def test_shard_lists_are_consistent():
    pts, tri = fa.make_icosphere(12, 0.1)
    part = fa.Mesh(pts, tri).partition(10)
    rowptr, col = part.ordered.topology()["rowptr"], part.ordered.topology()["col"]
    for world, k in ((1, 2), (2, 1), (5, 3)):
        sh = [fa.Shard(part, r, world, k) for r in range(world)]
        assert sum(s.n_own for s in sh) == part.ordered.V
        for s in sh:
            assert np.array_equal(s.loc2glob[:s.n_own], np.arange(s.own_begin, s.own_end))
            g = s.loc2glob[s.n_own:]
            assert np.all(np.diff(g.astype(np.int64)) > 0) and not ((g >= s.own_begin) & (g < s.own_end)).any()
            # ghosts = exactly the vertices within k edges of the owned range (BFS in numpy)
            dist_ = np.full(part.ordered.V, -1)
            dist_[s.own_begin:s.own_end] = 0
            frontier = np.arange(s.own_begin, s.own_end)
            for ring in range(1, k + 1):
                nb = np.unique(np.concatenate([col[rowptr[v]:rowptr[v + 1]] for v in frontier])) if len(frontier) else frontier
                frontier = nb[dist_[nb] < 0]
                dist_[frontier] = ring
            assert np.array_equal(g, np.nonzero(dist_ > 0)[0])
            off = 0
            for q in range(world):
                ids = s.loc2glob[s.send_idx[off:off + s.send_counts[q]]]
                off += s.send_counts[q]
                gq = sh[q].loc2glob[sh[q].n_own:]
                assert np.array_equal(ids, gq[(gq >= s.own_begin) & (gq < s.own_end)])
                assert s.send_counts[q] == sh[q].recv_counts[s.rank]
    with pytest.raises(fa.FemError):
        fa.Shard(part, 0, 11, 1)               # more ranks than partitions


@pytest.mark.gpu
@pytest.mark.parametrize("world,krings,precision,ksteps", [(4, 1, "f64", 0), (4, 3, "f64", 0), (3, 8, "f32", 0),
                                                           (4, 4, "f64", 2), (3, 8, "f64", 4), (2, 8, "f32", 4)])
def test_gpu_shards_loopback(world, krings, precision, ksteps):
    """several shards on one card, exchanged in-process: HIP shard plans + pack kernel"""
    import torch
    from fem_amd.dist import HipShardStepper
    nu, nparts, steps = 24, 40, 20
    part, (n_ref, c_ref) = reference_run(nu, nparts, steps)
    om = part.ordered
    om.init_stats(fa.PREC_F64)
    dt = 0.1 * om.stats()["maxEdgeLen"] ** 2
    model = fa.default_model()
    n0, c0 = fa.initial_condition(om.V, 3)
    shards = [fa.Shard(part, r, world, krings) for r in range(world)]
    steppers = [HipShardStepper(om, s, precision, model, n0[s.loc2glob], c0[s.loc2glob], part=part, ksteps=ksteps)
                for s in shards]
    budget = krings
    chunk = max(1, ksteps)
    assert krings % chunk == 0 and steps % chunk == 0
    for _ in range(steps // chunk):
        if budget == 0:
            sends = [st.pack(None).clone() for st in steppers]
            torch.cuda.synchronize()
            for r, st in enumerate(steppers):
                ghost, pos = st.ghost_view(), 0
                for q in range(world):
                    cnt = int(shards[r].recv_counts[q])
                    off = int(shards[q].send_counts[:r].sum())
                    ghost[pos:pos + cnt] = sends[q][off:off + cnt]
                    pos += cnt
            budget = krings
        for st in steppers:
            st.step(chunk, dt, model, None)
        budget -= chunk
    tol = 1e-11 if precision == "f64" else 1e-5
    for s, st in zip(shards, steppers):
        n, c = st.get_owned()
        parity(n, c, n_ref[s.own_begin:s.own_end], c_ref[s.own_begin:s.own_end], tol)


@pytest.mark.gpu
@pytest.mark.parametrize("nu,nparts,world,krings,fuse,precision,overlap", [
    (32, 96, 2, 4, 2, "f64", "1"),    # two groups per exchange: [all] then [boundary | exchange || interior]
    (32, 96, 2, 4, 2, "f64", "split"),  # exchange "not yet arrived": interior first, wait, then boundary + ghost
    (32, 96, 3, 3, 3, "f64", "split"),  # one group per exchange: interior, wait, boundary, exchange
    (32, 96, 3, 3, 3, "f64", "1"),    # one group per exchange, exchange already arrived: wait, boundary, exchange, interior
    (16, 12, 2, 6, 4, "f64", "1"),    # groups of 4 + 2 steps; every tile is a boundary tile
    (32, 96, 2, 4, 2, "f64", "0"),    # FEM_OVERLAP=0: plain schedule
    (16, 12, 2, 2, 0, "f64", "1"),    # sliced-ELL plans: plain schedule
])
def test_gpu_ranks_gloo(tmp_path, nu, nparts, world, krings, fuse, precision, overlap, monkeypatch):
    """ShardedRun with the HIP stepper as several processes sharing the card (exchange staged via
    gloo): the overlapped schedule over the [interior | boundary | ghost] tile classes"""
    steps = 11
    part, (n_ref, c_ref) = reference_run(nu, nparts, steps)
    monkeypatch.setenv("FEM_OVERLAP", overlap)
    res = launch(world, tmp_path, backend="hip", sphere=nu, nparts=nparts, krings=krings, steps=steps,
                 precision=precision, fuse=fuse)
    for r in res:
        b, e = int(r["own_begin"]), int(r["own_end"])
        parity(r["n"], r["c"], n_ref[b:e], c_ref[b:e], 1e-11)
        if overlap != "0" and fuse > 0:
            nI, nB, nG = r["classes"]
            assert nB > 0 and nG > 0 and (nI > 0 or nu < 32)      # the big case exercises all three classes
        else:
            assert tuple(r["classes"]) == (-1, -1, -1)


@pytest.mark.gpu
def test_gpu_tile_classes():
    """the class ranges cover the tiles; a whole-mesh plan is all interior; launching the classes
    one by one equals one launch of everything"""
    nu, nparts = 24, 40
    part, _ = reference_run(nu, nparts, 0)
    om = part.ordered
    om.init_stats(fa.PREC_F64)
    dt = 0.1 * om.stats()["maxEdgeLen"] ** 2
    n0, c0 = fa.initial_condition(om.V, 3)
    whole = fa.Plan(om, fa.PREC_F64, part=part, ksteps=3)
    assert whole.tile_classes() == (whole.tile_info()["ntiles"], 0, 0)
    shard = fa.Shard(part, 1, 3, 3)
    outs = []
    for split in (False, True):
        plan = fa.Plan(om, fa.PREC_F64, shard=shard, part=part, ksteps=3)
        nI, nB, nG = plan.tile_classes()
        assert nI + nB + nG == plan.tile_info()["ntiles"] and nB > 0 and nG > 0
        plan.set_state(n0[shard.loc2glob], c0[shard.loc2glob])
        if split:
            plan.launch_tiles(nI + nB, nG, 3, dt)
            plan.launch_tiles(0, nI, 3, dt)
            plan.launch_tiles(nI, nB, 3, dt)
            plan.flip()
        else:
            plan.run(3, dt)
        outs.append(plan.get_state())
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    with pytest.raises(fa.FemError):
        plan.launch_tiles(0, nI + nB + nG + 1, 3, dt)
    with pytest.raises(fa.FemError):
        plan.launch_tiles(0, 1, 4, dt)                      # more steps than the plan fuses
    with pytest.raises(fa.FemError):
        fa.Plan(om, fa.PREC_F64).tile_classes()             # sliced-ELL plan


@pytest.mark.gpu
@pytest.mark.parametrize("fuse,krings,overlap", [(0, 4, "1"), (5, 10, "1"), (5, 5, "split")])
def test_rccl_world_of_one(tmp_path, fuse, krings, overlap, monkeypatch):
    """the RCCL code path itself on the one GPU a test box has: backend nccl, world size 1.  The shard
    is the whole mesh (no ghosts), so the all-to-all moves nothing -- but the calls are the real ones:
    torch tensors over library memory handed to RCCL, the asynchronous work handle, is_completed(),
    wait() on the compute stream, the barrier with device_ids, every branch of the overlapped schedule"""
    monkeypatch.setenv("FEM_OVERLAP", overlap)
    nu, nparts, steps = 16, 8, 23
    part, (n_ref, c_ref) = reference_run(nu, nparts, steps)
    res = launch(1, tmp_path, backend="hip", sphere=nu, nparts=nparts, krings=krings, steps=steps, precision="f64",
                 fuse=fuse, pg="nccl")[0]
    assert int(res["own_begin"]) == 0 and int(res["own_end"]) == len(n_ref)
    parity(res["n"], res["c"], n_ref, c_ref, 1e-11)


@pytest.mark.gpu
def test_rccl_writes_into_library_memory():
    """real bytes through RCCL into the state buffer libfem_hip.so owns (the halo exchange's data path),
    with the rank as its own peer: tests/rccl_selfcopy_worker.py"""
    _launches[0] += 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + (os.getpid() * 16 + _launches[0]) % 2000),
           os.path.join(ROOT, "tests", "rccl_selfcopy_worker.py")]
    res = subprocess.run(cmd, timeout=150, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True)
    assert res.returncode == 0 and "selfcopy ok" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


@pytest.mark.gpu
def test_bench_survives_rccl_bringup_failure():
    """bench.py as the driver launches it for N > 1 (backend nccl), but with both ranks on ONE GPU
    (FEM_BENCH_SHARE_GPU=1): RCCL refuses duplicate devices at its first collective, the ranks agree
    over the gloo control group to stage the exchange through the host, and the bench still reports
    -- saying so in config.parallelism."""
    import json
    _launches[0] += 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + (os.getpid() * 16 + _launches[0]) % 2000),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "2", "--workload", "sphere10k"]
    env = dict(os.environ, FEM_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, check=True, timeout=150, env=env, capture_output=True, text=True)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    assert "STAGED THROUGH THE HOST" in out["config"]["parallelism"]


@pytest.mark.gpu
@pytest.mark.parametrize("steps,warmup", [(20, 5), (100, 2)])
def test_bench_self_launch_verifies_at_two_ranks(steps, warmup):
    """`python bench.py --gpus 2` from a bare shell (no torchrun, WORLD_SIZE unset): the parent starts the two ranks
    itself; the line carries `verify` of the gathered owned rows against the oracle (the state the timed region left
    for <= 64 steps, a pre-run on the same plans for longer runs) and `config.sustained`"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "4"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload", "sphere10k",
           "--steps", str(steps), "--warmup", str(warmup), "--sustained-steps", "60"]
    res = subprocess.run(cmd, timeout=300, env=env, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-2000:]
    out = json.loads(lines[0])
    v = out["verify"]
    assert out["n_gpus"] == 2 and v["ok"] and v["state_changed"]
    assert v["steps"] == (steps + warmup if steps + warmup <= 64 else 25)
    assert v["rel_l2_n_deviation"] <= 1e-5 and v["rel_l2_n"] <= 1e-5
    assert out["config"]["sustained"]["steps"] == 60 and out["config"]["sustained"]["value"] > 0
    assert out["roofline"]["vertices_per_launch"] == "rank 0's owned rows"


@pytest.mark.gpu
def test_bench_on_a_scan_file_with_pathologies(tmp_path):
    """`bench.py --mesh FILE.ply --subdivide 1` (config 4 with a file instead of the synthetic stand-in): a scanner-style
    binary PLY with open boundaries, unreferenced points and a non-manifold vertex goes through read -> upsample ->
    partition -> fused fp32 plan, and the line verifies against the oracle"""
    import json
    from helpers import pathological_scan, write_binary_ply
    pts, tri = pathological_scan(40)
    path = str(tmp_path / "scan.ply")
    write_binary_ply(path, pts, tri)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "4"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mesh", path, "--subdivide", "1", "--steps", "20", "--warmup", "5",
           "--sustained-steps", "0", "--no-cpu-baseline"]
    res = subprocess.run(cmd, timeout=300, env=env, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["data"] == "file scan.ply" and out["verify"]["ok"] and out["verify"]["steps"] == 25
    assert out["config"]["mesh"]["valence_histogram"][0] > 10          # vertices without faces were kept


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_bench_config5_on_two_ranks(precision):
    """BASELINE config 5 sharded (`bench.py --workload deform250k --gpus 2`, gloo rehearsal on one GPU): every rank
    deforms and re-assembles the rows of its local vertices, ghost positions deform locally, the state is exchanged
    every 4 iterations; the gathered owned rows are checked against the oracle re-assembling the deformed mesh every
    iteration (fp32: the oracle's mixed-precision assembly, 2e-5; fp64: 1e-11)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "4"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload", "deform250k",
           "--steps", "9", "--warmup", "2", "--sustained-steps", "0", "--precision", precision]
    res = subprocess.run(cmd, timeout=400, env=env, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    v = out["verify"]
    assert out["n_gpus"] == 2 and v["ok"] and v["steps"] == 11, v
    assert max(v["rel_l2_n"], v["rel_l2_c"]) <= (2e-5 if precision == "f32" else 1e-11)
    assert out["roofline"]["kernel"] == "assemble_step_sell_kernel"
