#!/usr/bin/env python3
"""bench.py -- DOF-iterations/s of the explicit time-step loop (BASELINE.json metric).

One "step" = one explicit Euler time step of the whole mesh (one `step` call in the
reference, src/FEM_common.cu:168-180; timing protocol of src/main.cpp:547-567: warm-up
steps, then back-to-back launches with no host sync inside, device-event timed).

  python bench.py --gpus 1 --steps K --warmup W [--workload sphere16m|sphere4m|sphere1m|sphere10k|scan8m|scan500k]
                  [--fuse k] [--mesh FILE.off|.ply [--subdivide L]]
  python bench.py --workload deform4m [--unfused]      config 5: re-assembly on the device every step

  python bench.py --gpus N ...        N > 1 from a bare shell: this process starts N fresh rank processes (one per GPU,
                                      RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), relays rank 0's line and exits non-zero
                                      if any rank does; it never touches the GPU itself.  Under torch.distributed.run
                                      (WORLD_SIZE already set) the process is a rank.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the byte accounting.
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WORKLOADS = {                       # class-I geodesic spheres, edge ~ 0.1
    "sphere10k": dict(nu=32, verts_per_part=197),        # config 1 size (P = 52)
    "sphere1m": dict(nu=320, verts_per_part=256),        # config 2
    "sphere4m": dict(nu=640, verts_per_part=256),
    "sphere16m": dict(nu=1280, verts_per_part=256),      # config 3 / north-star roofline size
    # config 4 stand-in (no scan data ships with the reference or this image): irregular valences
    # 3..16, local high-valence fans, radial noise; 8.1M vertices
    "scan8m": dict(nu=900, verts_per_part=256, scanned=True),
    "scan500k": dict(nu=224, verts_per_part=256, scanned=True),
    # config 5: deforming mesh, the operator re-assembled on the device every time step (one iteration =
    # deform kernel + fused assembly-and-step kernel); N > 1: shards with 4 ghost rings, state exchanged every 4 iterations
    "deform4m": dict(nu=640, verts_per_part=256, deform=True),
    "deform250k": dict(nu=160, verts_per_part=256, deform=True),
}
ALG_BYTES_PER_ELEMENT_ASSEMBLY = 68  # 54 B per element + 14 B of per-vertex output
ALG_BYTES_PER_VERTEX_STEP = 168     # V*24 + nnz*8 + F*48 on a closed genus-0 mesh
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_thread_share(local_world):
    """host worker threads one rank may use when `local_world` ranks share this host: the cores this process may
    run on, divided by the ranks, within [2, 16] (16 = the library's own default cap)"""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    return max(2, min(16, cores // max(1, local_world)))


def launch_ranks(nranks, cmd, env=None, grace=10.0):
    """One fresh process per rank, the way the reference's harness is one process that times itself
    (src/main.cpp:547-575) and the driver's `python bench.py --gpus N` expects: `cmd` is started `nranks` times with
    RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT (a free port) set.
    Rank 0's stdout is relayed line by line to this process's stdout, the other ranks' stdout goes to stderr.
    Returns 0 when every rank exited 0; as soon as one rank fails, the others (exact PIDs) get SIGTERM, then SIGKILL
    after `grace` seconds, and the first non-zero status is returned.  The caller must not have touched the GPU:
    nothing here does, and no process is replaced (no exec of an initialised process)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("FEM_HOST_THREADS", str(host_thread_share(nranks)))
    procs = []
    for r in range(nranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr,
                                      text=True, bufsize=1))

    def relay(pipe):
        for line in pipe:
            sys.stdout.write(line)
            sys.stdout.flush()
    t = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    t.start()

    def forward(signum, _frame):
        for p in procs:
            if p.poll() is None:
                p.send_signal(signum)
    old = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGINT)}
    status = 0
    try:
        live = set(range(nranks))
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 128 - rc
                    print("bench.py: rank %d exited with status %d; stopping the other ranks" % (r, rc), file=sys.stderr)
                    for q in live:
                        procs[q].terminate()
                    deadline = time.monotonic() + grace
                    while any(procs[q].poll() is None for q in live) and time.monotonic() < deadline:
                        time.sleep(0.05)
                    for q in live:
                        if procs[q].poll() is None:
                            procs[q].kill()
            time.sleep(0.02)
        for p in procs:
            p.wait()
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    t.join(timeout=5.0)
    return status


def cpu_baseline(seconds_budget=12.0):
    """the oracle (C port of CPUrun, no half-edge chasing, no per-step allocation) timed on this
    host on a bounded sample of the workload: the nu=320 sphere (1 024 002 vertices), as many steps
    as fit the budget; once on 1 core and once with OpenMP on the box's CPU share"""
    import fem_amd as fa
    import oracle_lib as ol
    nu = 320
    pts, tri = fa.make_icosphere(nu, 0.1)
    topo = ol.Topology(len(pts), tri)
    st = ol.Stats(topo, pts, ol.F64)
    n, c = ol.initial_condition(topo.V, 1)
    dt = 0.1 * st.maxEdgeLen ** 2
    model = fa.default_model().as_array()
    out = {}
    ncores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    for threads in (1, ncores):
        used = ol.set_threads(threads)
        t0 = time.perf_counter()
        st.step(n, c, dt, 1, model)
        one = time.perf_counter() - t0
        steps = max(2, min(400, int(seconds_budget / max(one, 1e-6))))
        t0 = time.perf_counter()
        st.step(n, c, dt, steps, model)
        el = time.perf_counter() - t0
        out[used] = (2.0 * topo.V * steps / el, steps, el)
    ol.set_threads(1)
    best = max(out, key=lambda k: out[k][0])
    return {"value": out[best][0], "unit": "DOF-iterations/s", "cores": best, "kind": "port",
            "single_core_value": out[1][0] if 1 in out else None,
            "sample": "oracle fo_step_f64 (C port of CPUrun, OpenMP), nu=320 icosphere V=%d, %d steps in %.1f s on %d "
                      "thread(s); 1 thread: %.3g DOF-it/s" % (topo.V, out[best][1], out[best][2], best,
                                                               out[1][0] if 1 in out else float("nan"))}


def cpu_baseline_deform(seconds_budget=12.0):
    """config 5 on the host: the oracle's assembly (fo_assemble_f64, serial C restatement of initMeshStats) of a
    freshly deformed nu=320 sphere + one fo_step_f64 per iteration, as many iterations as fit the budget"""
    import numpy as np
    import fem_amd as fa
    import oracle_lib as ol
    pts, tri = fa.make_icosphere(320, 0.1)
    topo = ol.Topology(len(pts), tri)
    n, c = ol.initial_condition(topo.V, 1)
    model = fa.default_model().as_array()
    threads = ol.set_threads(int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1))
    t0 = time.perf_counter()
    iters = 0
    while iters < 2 or (time.perf_counter() - t0 < seconds_budget and iters < 50):
        s = 1 + 0.01 * np.sin(1.5708 * pts[:, 0] + 0.01 * iters) * np.cos(1.5708 * pts[:, 1] - 0.005 * iters) * np.sin(1.5708 * pts[:, 2] + 0.0025 * iters)
        st = ol.Stats(topo, pts * s[:, None], ol.F64)
        n, c = st.step(n, c, 0.08 * st.maxEdgeLen ** 2, 1, model)
        iters += 1
    el = time.perf_counter() - t0
    ol.set_threads(1)
    return {"value": 2.0 * topo.V * iters / el, "unit": "DOF-iterations/s", "cores": threads, "kind": "port",
            "elements_per_s": topo.F * iters / el,
            "sample": "oracle fo_assemble_f64 (serial) + fo_step_f64 (OpenMP, %d threads) per iteration on a re-deformed nu=320 "
                      "icosphere V=%d, %d iterations in %.1f s" % (threads, topo.V, iters, el)}


def verify_deform(runner, state, total_steps):
    """config 5: the oracle deforms the ordered mesh with the same law, re-assembles it on the CPU (mixed precision =
    the reference's Vec3f geometry for the fp32 plan) and steps, every iteration; compared with the state the
    device loop left.  Tolerance 2e-5 (fp32 geometry round-off), 1e-11 for fp64."""
    import numpy as np
    import fem_amd as fa
    import oracle_lib as ol
    t0 = time.perf_counter()
    om = runner.part.ordered
    pts = om.points
    topo = ol.Topology(om.V, om.tri)
    n, c = fa.initial_condition(om.V, runner.seed)
    f32 = runner.precision == "f32"
    if f32:
        bn, bc = runner.model.nMax, runner.model.nMax / (1.0 + runner.model.nMax)
        n = bn + (n - bn).astype(np.float32).astype(np.float64)
        c = bc + (c - bc).astype(np.float32).astype(np.float64)
    threads = ol.set_threads(int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1))
    for i in range(total_steps):
        ph = runner.dphase * i
        s = 1 + runner.amp * np.sin(runner.q * pts[:, 0] + ph) * np.cos(runner.q * pts[:, 1] - 0.5 * ph) * np.sin(runner.q * pts[:, 2] + 0.25 * ph)
        st = ol.Stats(topo, pts * s[:, None], ol.MIXED if f32 else ol.F64)
        n, c = st.step(n, c, runner.dt, 1, runner.model.as_array())
    ol.set_threads(1)
    g = state

    def rel(a, b):
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    return {"rel_l2_n": rel(g[0], n), "rel_l2_c": rel(g[1], c), "steps": total_steps, "tolerance": 2e-5 if f32 else 1e-11,
            "oracle": "fo_assemble_%s of the re-deformed mesh + fo_step per iteration (oracle/, %d threads), %.1f s"
                      % ("mixed" if f32 else "f64", threads, time.perf_counter() - t0)}


def _rounded_ic(runner, V, precision):
    """the initial condition as the plan holds it: fp32 plans store fp32 deviations from the homogeneous state
    (fem_plan_set_state), so the oracle starts from those values"""
    import numpy as np
    import fem_amd as fa
    n0, c0 = fa.initial_condition(V, runner.seed)
    if precision == "f32":
        bn, bc = runner.model.nMax, runner.model.nMax / (1.0 + runner.model.nMax)
        n0 = bn + (n0 - bn).astype(np.float32).astype(np.float64)
        c0 = bc + (c0 - bc).astype(np.float32).astype(np.float64)
    return n0, c0


def oracle_reference(runner, total_steps, precision):
    """The check the reference's caller makes after its warm-up steps (GPUrun, main.cpp:492-541: the CPU path run
    for the same step count, then per-vertex differences) -- outside the timed region: the oracle's all-f64 CPUrun
    restatement (OpenMP) from the same initial condition on the same ordered mesh for `total_steps` time steps."""
    import oracle_lib as ol
    t0 = time.perf_counter()
    om = runner.part.ordered
    topo = ol.Topology(om.V, om.tri)
    st = ol.Stats(topo, om.points, ol.F64)
    n0, c0 = _rounded_ic(runner, om.V, precision)
    # (a launcher may have set OMP_NUM_THREADS=1 for every rank; the check runs on rank 0 alone, after the timed region,
    #  while the other ranks wait: it may use the host's cores)
    threads = ol.set_threads(max(int(os.environ.get("OMP_NUM_THREADS", "0") or 0), min(16, os.cpu_count() or 1)))
    n_ref, c_ref = st.step(n0, c0, runner.dt, total_steps, runner.model.as_array())
    ol.set_threads(1)
    return {"n": n_ref, "c": c_ref, "n0": n0, "steps": total_steps, "ic_precision": precision,
            "oracle": "fo_step_f64 (oracle/, %d threads), same IC and ordered mesh, %.1f s" % (threads, time.perf_counter() - t0)}


# fp32 / fp64 bars: relative L2 of the fields themselves (the north star's wording, <= 1e-5 fp32) AND relative to the
# DEVIATION from the homogeneous state, which is what the steps compute (||n - nMax|| ~ 0.003 ||n||): a 0.3 % error of
# the pattern would pass the first bar alone
TOL = {"f32": (1e-5, 1e-5), "f64": (1e-11, 1e-9),
       # the reference's own mix (fp64 state / Laplacian / update, fp32 gradient branch): the fp32 bar on the deviation
       # (observed at 16.4M: fields 6e-11, deviation 1.7e-9)
       "mixed": (1e-9, 1e-7)}


def compare_with_oracle(ref, state, model, precision):
    """`state` = (n, c) of the whole ordered mesh as the device left it after ref["steps"] steps"""
    import numpy as np
    n, c = state
    n_ref, c_ref = ref["n"], ref["c"]
    bn, bc = model.nMax, model.nMax / (1.0 + model.nMax)

    def rel(a, b):
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    tol, tol_dev = TOL[precision]
    v = {"rel_l2_n": rel(n, n_ref), "rel_l2_c": rel(c, c_ref),
         "rel_l2_n_deviation": float(np.linalg.norm(n - n_ref) / max(float(np.linalg.norm(n_ref - bn)), 1e-300)),
         "rel_l2_c_deviation": float(np.linalg.norm(c - c_ref) / max(float(np.linalg.norm(c_ref - bc)), 1e-300)),
         "max_abs_n": float(np.abs(n - n_ref).max()), "steps": ref["steps"], "tolerance": tol, "tolerance_deviation": tol_dev,
         "state_changed": bool(np.abs(n - ref["n0"]).max() > 0),   # the device state left the initial condition: the kernel ran
         "oracle": ref["oracle"]}
    v["ok"] = bool(max(v["rel_l2_n"], v["rel_l2_c"]) <= tol and
                   max(v["rel_l2_n_deviation"], v["rel_l2_c_deviation"]) <= tol_dev and v["state_changed"])
    return v


def gather_owned(runner, world, rank, group):
    """N > 1: the owned rows of every rank on rank 0 (over the gloo control group when there is one), as whole-mesh
    arrays in the ordered numbering; None on the other ranks"""
    import numpy as np
    import torch.distributed as dist
    n, c = runner.get_owned()
    got = [None] * world if rank == 0 else None
    dist.gather_object((int(runner.shard.own_begin), int(runner.shard.own_end), n, c), got, dst=0, group=group)
    if rank != 0:
        return None
    V = runner.part.ordered.V
    fn, fc, covered = np.full(V, np.nan), np.full(V, np.nan), 0
    for b, e, rn, rc in got:
        fn[b:e], fc[b:e] = rn, rc
        covered += e - b
    if covered != V:
        raise SystemExit("bench.py: the ranks' owned ranges cover %d of %d vertices" % (covered, V))
    return fn, fc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("FEM_BENCH_WORKLOAD", "sphere16m"), choices=sorted(WORKLOADS))
    ap.add_argument("--mesh", default=None, help="a real .off/.ply surface instead of a synthetic workload (config 4 with real scans)")
    ap.add_argument("--subdivide", type=int, default=0, help="rounds of 1-to-4 upsampling of --mesh")
    ap.add_argument("--precision", default="f32", choices=["f32", "f64", "mixed"],
                    help="mixed = fp64 plan with the reference's own mix: fp32 gradient branch (fused plans)")
    ap.add_argument("--fuse", type=int, default=None,
                    help="time steps fused per launch (0 = single-step sliced-ELL kernel); default FEM_FUSE_STEPS, "
                         "else the measured best: 7 on the spheres, 5 on the irregular scanned-like meshes")
    ap.add_argument("--unfused", action="store_true", help="deform workloads: assembly kernel, then step kernel (two kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the run's result")
    ap.add_argument("--no-variants", action="store_true",
                    help="N=1 fp32 fused runs: skip config.variants (the same mesh on the fp64 plan, in the reference's precision mix and on four-float records)")
    ap.add_argument("--verify-steps", type=int, default=25,
                    help="runs longer than 64 steps: this many steps are taken and checked against the oracle BEFORE the "
                         "timed region on the same plan (the state is reset afterwards)")
    ap.add_argument("--sustained-steps", type=int, default=1000,
                    help="the same kernel timed over this many further steps after the timed region (config.sustained); 0 = off")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal only (ranks may share a GPU; halo staged through the host)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: this process becomes the launcher and never touches the GPU
        raise SystemExit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    if args.precision == "mixed" and (WORKLOADS[args.workload].get("deform") or args.fuse == 0):
        raise SystemExit("bench.py: --precision mixed is a fused plan (FEM_PLAN_MIXED_RECORDS)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 and "FEM_HOST_THREADS" not in os.environ:
        # every rank reads, builds the topology of and partitions the same mesh (deterministic, so no broadcast is
        # needed): the ranks of one host share its cores instead of each taking the single-process default of 16
        os.environ["FEM_HOST_THREADS"] = str(host_thread_share(int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))))

    import numpy as np
    import torch
    import torch.distributed as dist

    import fem_amd as fa
    from fem_amd import dist as fdist

    share = args.dist_backend == "gloo" or os.environ.get("FEM_BENCH_SHARE_GPU") == "1"   # rehearsals only
    local_dev = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(local_dev)
    control = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            # RCCL communicator comes up at the first collective (ShardedRun's priming exchange);
            # the gloo control group lets all ranks agree on the host-staged fallback if it cannot
            dist.init_process_group("nccl")
            try:
                control = dist.new_group(backend="gloo")
            except Exception as ex:                      # noqa: BLE001  (no gloo transport: no fallback, RCCL only)
                print("bench.py: no gloo control group (%s); continuing with RCCL only" % type(ex).__name__, file=sys.stderr)
        else:
            dist.init_process_group("gloo")

    device_token = [None]

    def rank_barrier(device_side=False):
        """device_side (the bracket that CLOSES a timed region): with RCCL up, one all-reduce of a device word over it
        (tens of microseconds over xGMI; the synchronize() that follows waits for it) instead of the gloo control group's
        host all-reduce (hundreds of microseconds at 8 ranks, a large part of a 20-step region)"""
        if world == 1:
            return
        if device_side and getattr(runner_box[0], "host_group", True) is None and (
                args.dist_backend == "nccl" or os.environ.get("FEM_BENCH_DEVICE_BARRIER") == "1"):
            if device_token[0] is None:
                device_token[0] = torch.zeros(1, device="cuda")
            dist.all_reduce(device_token[0])                 # default group: RCCL (gloo in rehearsals that force it)
            return
        if control is not None:
            dist.all_reduce(torch.zeros(1), group=control)
        elif args.dist_backend == "nccl":
            dist.barrier(device_ids=[local_dev])
        else:
            dist.barrier()

    runner_box = [None]                                      # set once the runner exists (rank_barrier reads its RCCL state)

    def max_over_ranks(values):
        if world == 1:
            return values
        staged = control if control is not None else None
        t = torch.tensor(values, dtype=torch.float64,
                         device="cuda" if args.dist_backend == "nccl" and staged is None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=staged)
        return [float(x) for x in t]

    wl = WORKLOADS[args.workload] if args.mesh is None else dict(nu=0, verts_per_part=256, scanned=True)
    if args.fuse is None:
        args.fuse = -1                                      # chosen below, once the mesh size is known
    t_host = time.perf_counter()
    scanned = wl.get("scanned", False)
    if args.mesh is not None:
        pts, tri, _ = fa.read_mesh(args.mesh)
        if args.subdivide > 0:
            pts, tri = fa.subdivide(pts, tri, args.subdivide)
    else:
        pts, tri = fa.make_scanned_like(wl["nu"], 0.1, seed=3) if scanned else fa.make_icosphere(wl["nu"], 0.1)
    mesh = fa.Mesh(pts, tri)
    V_total, n_faces = mesh.V, mesh.F
    dt_rule = "stable" if scanned else "reference"
    # fused tiles: partition size such that a partition + (fuse-1) rings fits the register tile
    # (the library gives regular meshes the 2048-row register tile; meshes with rows wider than 6 entries and
    # plans below 1.5M rows -- kLargeMeshRows in device.hip, per rank -- get the 1536-row one)
    small = scanned or V_total // world < 1500000
    if small and not scanned and world == 1 and args.precision == "f32" and 192 <= V_total // 975 <= 2400:
        # ... except whole meshes of 192..2400 such tiles: fem_plan_run runs those in persistent launches (device.hip:
        # persistRule), which do not need many small tiles to fill the workgroup slots
        small = False
    # tiny meshes (config 1: 10k vertices) cannot fill the chip with tiles of several hundred rows: they take the deepest
    # fusion and ~ 160 small tiles instead (same-box sweep on sphere10k: 13.3 G with 13 tiles at k = 6, 22.1 G with 160 tiles
    # of 64 rows at k = 8; 40-row tiles 15.8 G)
    tiny = not scanned and V_total // world < 150000
    if args.fuse < 0:
        args.fuse = int(os.environ.get("FEM_FUSE_STEPS", "5" if scanned else ("8" if tiny else ("6" if small else "7"))))
    tile_n0 = int(os.environ.get("FEM_TILE_N0", ({4: "950", 5: "820" if scanned else "880", 6: "760", 7: "650"} if small else
                                                 {5: "1200", 6: "1050", 7: "975", 8: "900"}).get(args.fuse, "820")))
    if tiny and "FEM_TILE_N0" not in os.environ:
        tile_n0 = min(tile_n0, max(64, V_total // world // 160))
    nparts = max(world, V_total // (tile_n0 if args.fuse > 0 else wl["verts_per_part"]))
    deform = wl.get("deform", False)
    if deform:
        args.fuse, nparts = 0, max(world, V_total // wl["verts_per_part"])
        if world > 1:       # ghost positions deform locally; only the state is exchanged, every FEM_HALO_RINGS iterations
            runner = fdist.ShardedDeformRun(mesh, nparts, rank, world, args.precision, fused=not args.unfused,
                                            control_group=control)
        else:
            runner = fdist.DeformRun(mesh, nparts, args.precision, fused=not args.unfused)
    elif world > 1:
        runner = fdist.ShardedRun(mesh, nparts, rank, world, args.precision, ksteps=args.fuse, dt_rule=dt_rule,
                                  control_group=control)
    else:
        runner = fdist.SingleRun(mesh, nparts, args.precision, ksteps=args.fuse, dt_rule=dt_rule)
    runner_box[0] = runner
    if world > 1:                   # the first all-reduce over RCCL sets up its channels: not inside a timed region
        rank_barrier(device_side=True)
        torch.cuda.synchronize()
    mesh_report = None
    if scanned and rank == 0:           # config 4: valence histogram and partition imbalance
        val = np.bincount(tri.ravel().astype(np.int64), minlength=V_total)
        sizes = np.diff(runner.part.node_parts.astype(np.int64))
        fsizes = np.diff(runner.part.element_parts.astype(np.int64))[:-1]
        mesh_report = {"valence_histogram": np.bincount(val).tolist(),
                       "partition_vertices_min_mean_max": [int(sizes.min()), float(sizes.mean()), int(sizes.max())],
                       "partition_interior_faces_min_mean_max": [int(fsizes.min()), float(fsizes.mean()), int(fsizes.max())],
                       "cut_faces_fraction": float((runner.part.element_parts[-1] - runner.part.element_parts[-2]) / len(tri))}
    del pts, tri, mesh
    t_host = time.perf_counter() - t_host
    staged_group = getattr(runner, "host_group", None)
    gather_group = control if control is not None else None

    stream = torch.cuda.current_stream()

    def whole_state(r):
        """(n, c) of the whole ordered mesh on rank 0 (None elsewhere)"""
        if world > 1:
            return gather_owned(r, world, rank, gather_group)
        return r.plan.get_state()

    def timed(r, steps, warmup):
        """`warmup` untimed steps, then `steps` steps between a barrier + synchronize on both sides (wall clock, the
        MAX over the ranks) and between two events on the launch stream (device time)"""
        r.run(warmup, stream)
        torch.cuda.synchronize()
        rank_barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        r.run(steps, stream)
        e1.record(stream)
        torch.cuda.synchronize()
        rank_barrier(device_side=True)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        return max_over_ranks([wall, e0.elapsed_time(e1)])

    def sustained_of(r, steps):
        wall, dev = timed(r, steps, 0)
        return {"steps": steps, "value": 2.0 * V_total * steps / wall, "ms_per_step": wall * 1e3 / steps,
                "device_ms_per_step": dev / steps}

    # ---- the check (outside every timed region).  Runs of <= 64 steps: the state the TIMED kernel left is compared
    # with the oracle after warmup + steps steps.  Longer runs: --verify-steps steps are taken on the same plan before
    # the timed region, downloaded, and the state is reset to the initial condition.
    total = args.warmup + args.steps
    check_steps, checked_state, check_kind = 0, None, None
    do_verify = not args.no_verify
    if do_verify and total > (32 if deform else 64):
        check_steps = min(args.verify_steps, 10) if deform else args.verify_steps
        runner.run(check_steps, stream)
        torch.cuda.synchronize()
        checked_state = whole_state(runner)
        runner.reset()
        check_kind = "%d steps taken before the timed region on the same plan; state reset afterwards" % check_steps

    wall, dev_ms = timed(runner, args.steps, args.warmup)
    if do_verify and check_steps == 0:
        check_steps = total
        checked_state = whole_state(runner)              # what the timed kernel left
        check_kind = "the state the timed region left (warmup + steps)"
    sustained, asm_ms = None, None
    if args.sustained_steps > 0 and args.sustained_steps != args.steps:
        sustained = sustained_of(runner, args.sustained_steps)
    if deform:
        # the dominant kernel alone (no deformation kernel between the launches), device events on its stream
        reps = max(10, min(200, args.steps))
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(stream)
        runner.run(reps, stream, deform=False)
        a1.record(stream)
        torch.cuda.synchronize()
        asm_ms = a0.elapsed_time(a1) / reps

    # ---- the same mesh on the reference's arithmetic type (fp64, src/common.h:7-11), in the reference's own precision
    # mix (fp64 state and Laplacian, fp32 gradient branch: FEM_common.cu:82-166) and on lossless four-float operator
    # records, measured after the headline with the same protocol (N = 1, fused fp32 runs only)
    variants, variant_states = None, {}
    if (world == 1 and not deform and not args.no_variants and args.precision == "f32" and args.fuse > 0
            and args.mesh is None):
        variants = {}
        n0r, c0r = _rounded_ic(runner, V_total, "f32")       # the headline's effective IC: one oracle run serves all
        for name, prec, packed in (("f64", "f64", None), ("mixed", "mixed", None), ("f32_four_float", "f32", "0")):
            old = os.environ.get("FEM_TILE_PACKED")
            if packed is not None:
                os.environ["FEM_TILE_PACKED"] = packed
            try:
                vr = fdist.SingleRun(None, nparts, prec, ksteps=args.fuse, part=runner.part, dt=runner.dt)
            finally:
                if packed is not None:
                    if old is None:
                        del os.environ["FEM_TILE_PACKED"]
                    else:
                        os.environ["FEM_TILE_PACKED"] = old
            vr.plan.set_state(n0r, c0r)
            if do_verify and total > 64:
                vr.run(check_steps, stream)
                torch.cuda.synchronize()
                variant_states[name] = (vr.plan.get_state(), prec)
                vr.plan.set_state(n0r, c0r)
            vw, vd = timed(vr, args.steps, args.warmup)
            if do_verify and total <= 64:
                variant_states[name] = (vr.plan.get_state(), prec)
            ti = vr.plan.tile_info()
            variants[name] = {"value": 2.0 * V_total * args.steps / vw, "ms_per_step": vw * 1e3 / args.steps,
                              "dtype": prec, "operator_records": ti["operator_records"],
                              "tile": "%dx%d k%d" % (ti["threads"], ti["rows_per_thread"], ti["ksteps"]),
                              "sustained": sustained_of(vr, args.sustained_steps) if args.sustained_steps > 0 else None}
            del vr

    # ---- the same plan taking SEMI-IMPLICIT steps (diffusion implicit by Chebyshev iteration on the fused tiles,
    # fem_plan_run_imex_fused; no counterpart in the reference): how far it advances the model per millisecond
    # against the explicit steps just timed.  Not checked here (no oracle for it at this size): tests/test_gpu_implicit.py
    # compares it with scipy direct solves at oracle sizes.
    semi_implicit = None
    if variants is not None and args.mesh is None and hasattr(runner.plan, "run_imex_fused"):
        # (the synthetic workloads, the irregular one included: there the explicit step is a fraction of the diffusion limit
        #  of the WORST row, which is what an implicit diffusion term is for)
        factor, tol_si, nsi = 30.0, 1e-4, 5
        state_keep = runner.plan.get_state()
        runner.plan.run_imex_fused(2, factor * runner.dt, runner.model, rel_tol=tol_si, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.plan.run_imex_fused(nsi, factor * runner.dt, runner.model, rel_tol=tol_si, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        si = (time.perf_counter() - t0) / nsi
        ns_, _ = runner.plan.get_state()
        runner.plan.set_state(*state_keep)
        explicit_ms = (sustained["ms_per_step"] if sustained else wall * 1e3 / args.steps)
        semi_implicit = {"dt_over_explicit_dt": factor, "tolerance": tol_si, "chebyshev_iterations": runner.plan.imex_fused_iterations(),
                         "ms_per_step": si * 1e3, "explicit_ms_per_step_same_plan": explicit_ms,
                         "simulated_time_per_ms_vs_these_explicit_steps": factor * explicit_ms / (si * 1e3),
                         "state_finite": bool(np.isfinite(ns_).all()),
                         "checked": "against scipy direct solves at oracle sizes (tests/test_gpu_implicit.py), not at this size"}

    verify_failed = False
    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        spl = getattr(runner, "steps_per_launch", 1)
        launches = runner.launch_count(args.steps) if hasattr(runner, "launch_count") else -(-args.steps // spl)
        if hasattr(runner, "kernel_for"):                   # short runs of a mid-sized plan go one launch per group
            runner.kernel_name = runner.kernel_for(args.steps)
        kernel_ms = dev_ms / launches                       # average launch duration, device events
        v_per_launch = runner.vertices_per_launch           # owned vertices this rank advances per launch
        achieved = ALG_BYTES_PER_VERTEX_STEP * v_per_launch * (args.steps / launches) / (kernel_ms * 1e-3) / 1e9
        if deform:        # the assembly is the dominant kernel: 68 B per element by SURVEY.md 8(d)
            kernel_ms = asm_ms
            # (a shard assembles the rows of its local vertices: its share of the elements is priced by its owned rows)
            achieved = ALG_BYTES_PER_ELEMENT_ASSEMBLY * n_faces * (v_per_launch / V_total) / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_source = runner.measured_traffic()
        out = {
            "metric": "DOF-iterations/s", "value": 2.0 * V_total * args.steps / wall, "unit": "DOF-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic" if args.mesh is None else "file " + os.path.basename(args.mesh),
            "config": {"workload": "%s: %s nu=%d, V=%d, F=%d, %d partitions, "
                                   "reaction-diffusion explicit Euler, seeded random IC"
                                   % (args.workload if args.mesh is None else os.path.basename(args.mesh) + " x4^%d" % args.subdivide,
                                      "surface read from file" if args.mesh is not None else
                                      "scanned-like irregular surface (flipped/fanned/noisy geodesic sphere)"
                                      if scanned else ("class-I geodesic icosphere, DEFORMED and re-assembled on the device every step "
                                                       "(%s)" % ("assembly kernel + step kernel" if args.unfused else "one fused assembly-and-step kernel")
                                                       if deform else "class-I geodesic icosphere"), wl["nu"], V_total, n_faces, nparts),
                       "elements_per_s": n_faces * args.steps / wall if deform else None,
                       "assembly_elements_per_s": n_faces / (asm_ms * 1e-3) if deform else None,
                       "vertex_steps_per_s": V_total * args.steps / wall, "host_setup_s": t_host,
                       "host_threads_per_rank": int(os.environ.get("FEM_HOST_THREADS", "0")) or None,
                       "tiles": runner.plan.tile_info() if getattr(runner, "plan", None) is not None else None,
                       "halo_rings": getattr(runner, "krings", None), "mesh": mesh_report, "sustained": sustained,
                       "variants": variants, "semi_implicit": semi_implicit,
                       "parallelism": "1 rank" if world == 1 else "%d ranks (%s), partition ranges + halo exchange%s" % (
                           world, "RCCL" if args.dist_backend == "nccl" else "gloo rehearsal, halo staged through the host",
                           "" if getattr(runner, "fallback_reason", None) is None else
                           " STAGED THROUGH THE HOST over gloo (RCCL did not come up: %s)" % runner.fallback_reason)},
            "roofline": {"bound": "hbm", "kernel": runner.kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("recorded earlier for exactly this configuration (%s) and kernel build (%s), not measured "
                                            "by this run: %s" % (runner.traffic_key(), fa.kernel_build(), traffic_source)) if traffic
                         else traffic_source,
                         "traffic_key": runner.traffic_key(), "kernel_build": fa.kernel_build(),
                         "traffic_rate": traffic / (kernel_ms * 1e-3) / 1e9 if traffic else None,
                         "traffic_frac": traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
                         "alg_bytes_per_vertex_step": None if deform else ALG_BYTES_PER_VERTEX_STEP,
                         "alg_bytes_per_element": ALG_BYTES_PER_ELEMENT_ASSEMBLY if deform else None, "avg_launch_ms": kernel_ms,
                         "time_steps_per_launch": args.steps / launches,
                         "vertices_per_launch": "rank 0's owned rows" if world > 1 else "all rows",
                         "layout_bytes_per_launch": runner.layout_bytes_per_launch},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_deform() if deform else cpu_baseline()
        if checked_state is not None:
            if deform:
                v = out["verify"] = verify_deform(runner, checked_state, check_steps)
                v["ok"] = bool(max(v["rel_l2_n"], v["rel_l2_c"]) <= v["tolerance"])
            else:
                ref = oracle_reference(runner, check_steps, args.precision)
                v = out["verify"] = compare_with_oracle(ref, checked_state, runner.model, args.precision)
                for name, (st, prec) in variant_states.items():
                    vv = variants[name]["verify"] = compare_with_oracle(ref, st, runner.model, prec)
                    # the fp64 plan started from the fp32-rounded IC the oracle used: its own bars apply
                    verify_failed = verify_failed or not vv["ok"]
            v["checked"] = check_kind
            verify_failed = verify_failed or not v["ok"]
        print(json.dumps(out), flush=True)
    if world > 1:
        rank_barrier()                                      # rank 0's oracle check is done: tear down together
        dist.destroy_process_group()
    if verify_failed:
        raise SystemExit("bench.py: the run's result differs from the oracle beyond the tolerance (see \"verify\")")


if __name__ == "__main__":
    main()
