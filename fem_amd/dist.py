"""Run drivers over the C-ABI: a whole mesh on one GPU (SingleRun) and a mesh sharded along
the partition ranges over the ranks of one node (ShardedRun, halo exchange through
torch.distributed).  Plumbing only: all arithmetic happens in libfem_hip.so."""
import json
import os

import numpy as np

import fem_amd as fa

_PROFILES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def _partition(mesh, nparts):
    """R0 stage by FEM_PARTITIONER: auto (default: rcb -- exactly equal parts, lowest cut and ring redundancy on the
    surfaces measured -- unless its cut says the coordinates mislead, then graph), rcb, graph, multilevel
    (METIS_PartMeshDual-style: multilevel k-way on the dual graph)"""
    pre = None
    if os.environ.get("FEM_WIDE_FIRST", "1") != "0":
        # irregular meshes: rows of more than 6 entries numbered first, so that they lead every part (the R2 stage keeps
        # the input order inside a part) and share waves in the fused kernel (fem_mesh_wide_rows_first)
        mesh, pre = mesh.wide_rows_first(6)
    part = mesh.partition(nparts, method=os.environ.get("FEM_PARTITIONER", "auto"))
    if pre is not None:
        part.perm_v = part.perm_v[pre]          # new position of each vertex of the mesh the caller passed
    return part


def _time_step(om, model, rule):
    """"reference": dt = 0.1 maxEdgeLen^2 (main.cpp:188; assumes near-uniform triangles);
    "stable": inside the explicit diffusion limit of the assembled operator (irregular meshes)"""
    st = om.stats()
    if rule == "stable":
        return 0.2 / (max(1.0, model.D) * float(np.abs(st["wVtx"]).max()))
    return 0.1 * st["maxEdgeLen"] ** 2


class SingleRun:
    """partition -> reorder -> assemble (host, once) -> native plan; seeded IC of main.cpp:178-186"""
    def __init__(self, mesh, nparts, precision="f32", seed=1, ksteps=0, dt_rule="reference", part=None, dt=None):
        """`part` (+ `dt`): a Partition whose ordered mesh already carries its f64 stats -- a second plan on the
        same mesh (bench.py's precision/format variants) repeats neither the partition nor the assembly"""
        self.ksteps = ksteps
        self.kernel_name = "tile_step_kernel" if ksteps > 0 else "sell_step_kernel"
        self.part = part if part is not None else _partition(mesh, nparts)
        om = self.part.ordered
        self.model = fa.default_model()
        if part is None or dt is None:
            om.init_stats(fa.PREC_F64)
            dt = _time_step(om, self.model, dt_rule)
        self.dt = dt
        self.seed = seed
        self.precision = precision
        # "mixed": an fp64 plan in the reference's own precision mix (fp32 gradient branch, FEM_PLAN_MIXED_RECORDS)
        self.plan = fa.Plan(om, fa.PREC_F32 if precision == "f32" else fa.PREC_F64, part=self.part, ksteps=ksteps,
                            mixed=precision == "mixed")
        n0, c0 = fa.initial_condition(om.V, seed)
        self.plan.set_state(n0, c0)
        self.V = om.V
        self.vertices_per_launch = om.V                         # per time step
        self.steps_per_launch = max(1, ksteps)
        sb = 4 if precision == "f32" else 8
        self.layout_bytes_per_launch = self.plan.operator_bytes() + 4 * sb * om.V   # operator + state r/w
        groups = self.plan.persist_groups() if ksteps > 0 else 0
        if groups > 0:          # mid-sized plans: several groups of k steps per launch, tiles handed over inside it
            self.kernel_name = "tile_step_persist_kernel"
            self.steps_per_launch = groups * ksteps
            self.layout_bytes_per_launch *= groups              # every group streams the operator and the state again

    def launch_count(self, nsteps):
        """launches run(nsteps) makes (bench.py's average launch duration): a run too short for persistent launches,
        or its remainder, goes one launch per group"""
        if self.ksteps <= 0:
            return int(nsteps)
        return self.plan.launch_count(nsteps)[0]

    def kernel_for(self, nsteps):
        """the kernel run(nsteps) spends its time in"""
        if self.ksteps <= 0:
            return self.kernel_name
        return "tile_step_persist_kernel" if self.plan.launch_count(nsteps)[1] > 0 else "tile_step_kernel"

    def run(self, nsteps, stream=None):
        self.plan.run(nsteps, self.dt, self.model, stream.cuda_stream if stream is not None else None)

    def reset(self):
        """back to the seeded initial condition (bench.py: after the steps taken for the oracle check)"""
        n0, c0 = fa.initial_condition(self.V, self.seed)
        self.plan.set_state(n0, c0)

    def traffic_key(self):
        """what a PMC pass is valid for: kernel, precision, rows, fused depth, register tile, partition count"""
        ti = self.plan.tile_info()
        return "%s:%s:%d:k%d:%dx%d:p%d%s" % (self.kernel_name, self.precision, self.V, ti["ksteps"], ti["threads"],
                                              ti["rows_per_thread"], self.part.n,
                                              {1: ":rec12", 2: ":rec20"}.get(self.plan.operator_format(), ""))

    def measured_traffic(self):
        return _recorded_traffic(self.traffic_key())


def _recorded_traffic(key):
    """(HBM bytes per launch, source tag) from the rocprofv3 PMC passes committed under profiles/ -- recorded on an earlier
    run of exactly this configuration (traffic_key) AND this kernel build (fem_kernel_build: a digest of the device
    sources), not measured by this process; (None, reason) when no pass was recorded for it or the kernels have
    changed since"""
    path = os.path.join(_PROFILES, "traffic.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        rec = json.load(f)
    hit = rec.get(key)
    if not isinstance(hit, dict):
        return None, None
    if hit.get("kernel_build") != fa.kernel_build():
        return None, "recorded for kernel build %s (%s); this library is %s" % (hit.get("kernel_build"), hit.get("source"), fa.kernel_build())
    return hit.get("hbm_bytes_per_launch"), hit.get("source")


class DeformRun:
    """BASELINE config 5: a deforming mesh, the operator re-assembled on the device every time step.
    One iteration = deform kernel (seeded smooth radial displacement, phase advancing per step) + the fused
    assembly + step kernel (fem_plan_assemble_step); `fused=False`: assembly kernel, then step kernel."""
    def __init__(self, mesh, nparts, precision="f32", seed=1, fused=True):
        self.kernel_name = "assemble_step_sell_kernel" if fused else "assemble_rows_sell_kernel"
        self.fused = fused
        self.part = _partition(mesh, nparts)
        om = self.part.ordered
        om.init_stats(fa.PREC_F64)
        self.model = fa.default_model()
        self.dt = 0.08 * om.stats()["maxEdgeLen"] ** 2          # inside the explicit limit for every deformed state
        self.precision = precision
        self.plan = fa.Plan(om, fa.PREC_F32 if precision == "f32" else fa.PREC_F64, part=self.part, ksteps=0, assemble=True)
        self.seed = seed
        n0, c0 = fa.initial_condition(om.V, seed)
        self.plan.set_state(n0, c0)
        self.V, self.F = om.V, om.F
        self.vertices_per_launch = om.V
        self.steps_per_launch = 1
        self.amp, self.q, self.dphase = 0.01, 2 * np.pi / 4.0, 0.01
        self.iteration = 0
        self.layout_bytes_per_launch = None

    def run(self, nsteps, stream=None, deform=True):
        s = stream.cuda_stream if stream is not None else None
        for _ in range(nsteps):
            if deform:
                self.plan.deform(self.amp, self.q, self.dphase * self.iteration, s)
            if self.fused:
                self.plan.assemble_step(self.dt, self.model, False, s)
            else:
                self.plan.assemble(s)
                self.plan.run(1, self.dt, self.model, s)
            self.iteration += 1

    def reset(self):
        n0, c0 = fa.initial_condition(self.V, self.seed)
        self.plan.set_state(n0, c0)
        self.iteration = 0

    def traffic_key(self):
        return "%s:%s:%d" % (self.kernel_name, self.precision, self.V)

    def measured_traffic(self):
        return _recorded_traffic(self.traffic_key())


class ShardedDeformRun:
    """BASELINE config 5 on shards: every rank deforms the vertices of its own rows (owned + k ghost rings; the
    deformation is a function of the base position, so ghost positions need no exchange), re-assembles the operator
    rows of its local vertices and takes the time step in one kernel (fem_plan_assemble_step), k iterations between
    two halo exchanges of the STATE (rows of ring j are exact for j <= k - s after s iterations, as in ShardedRun)."""
    def __init__(self, mesh, nparts, rank, world, precision="f32", seed=1, krings=4, fused=True, control_group=None):
        self.kernel_name = "assemble_step_sell_kernel" if fused else "assemble_rows_sell_kernel"
        self.fused = fused
        self.part = _partition(mesh, nparts)
        om = self.part.ordered
        self.model = fa.default_model()
        self.precision = precision
        self.krings = int(os.environ.get("FEM_HALO_RINGS", str(krings)))
        self.shard = fa.Shard(self.part, rank, world, self.krings)
        om.init_stats_shard(fa.PREC_F64, self.shard)
        self.dt = 0.08 * om.stats()["maxEdgeLen"] ** 2           # maxEdgeLen is the whole mesh's on every rank
        self.seed = seed
        n0, c0 = fa.initial_condition(om.V, seed)
        l2g = self.shard.loc2glob
        self.stepper = HipShardStepper(om, self.shard, precision, self.model, n0[l2g], c0[l2g], part=self.part, ksteps=0,
                                       assemble=True)
        self.plan = self.stepper.plan
        self.V, self.F = om.V, om.F
        self.vertices_per_launch = self.shard.n_own
        self.steps_per_launch = 1
        self.amp, self.q, self.dphase = 0.01, 2 * np.pi / 4.0, 0.01
        self.iteration = 0
        self.budget = self.krings
        self.layout_bytes_per_launch = None
        self.host_group, self.fallback_reason = None, None
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and world > 1:
            err = None
            try:
                exchange_halo(self.stepper, self.shard, None)      # brings the communicator up outside the timed region
            except Exception as ex:                                # noqa: BLE001
                if control_group is None:
                    raise
                err = ex
            if control_group is not None:
                import torch
                failed = torch.tensor([0 if err is None else 1])
                dist.all_reduce(failed, op=dist.ReduceOp.MAX, group=control_group)
                if int(failed.item()):
                    self.host_group = control_group
                    self.fallback_reason = ("%s: %s" % (type(err).__name__, str(err).strip().splitlines()[0][:160])
                                            if err is not None else "RCCL failed on another rank")
                    exchange_halo(self.stepper, self.shard, None, host_group=control_group)

    def run(self, nsteps, stream=None, deform=True):
        s = stream.cuda_stream if stream is not None else None
        for _ in range(nsteps):
            if self.budget == 0:
                exchange_halo(self.stepper, self.shard, stream, host_group=self.host_group)
                self.budget = self.krings
            if deform:
                self.plan.deform(self.amp, self.q, self.dphase * self.iteration, s)
            if self.fused:
                self.plan.assemble_step(self.dt, self.model, False, s)
            else:
                self.plan.assemble(s)
                self.plan.run(1, self.dt, self.model, s)
            self.iteration += 1
            self.budget -= 1

    def get_owned(self):
        return self.stepper.get_owned()

    def reset(self):
        n0, c0 = fa.initial_condition(self.V, self.seed)
        l2g = self.shard.loc2glob
        self.plan.set_state(n0[l2g], c0[l2g])
        self.iteration, self.budget = 0, self.krings

    def traffic_key(self):
        return "%s:%s:shard %d of %d" % (self.kernel_name, self.precision, self.shard.rank, self.shard.nranks)

    def measured_traffic(self):
        return None, None


class _DevView:
    """zero-copy view of device memory owned by the library, for torch.as_tensor"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


class HipShardStepper:
    """one rank's sharded plan on its GPU"""

    def __init__(self, ordered, shard, precision, model, n_loc, c_loc, part=None, ksteps=0, assemble=False):
        import torch
        self.torch = torch
        self.shard = shard
        self.plan = fa.Plan(ordered, fa.PREC_F32 if precision == "f32" else fa.PREC_F64, model, shard=shard,
                            part=part, ksteps=ksteps, assemble=assemble, mixed=precision == "mixed")
        self.plan.set_state(n_loc, c_loc)
        self.dtype = torch.float32 if precision == "f32" else torch.float64
        self.typestr = "<f4" if precision == "f32" else "<f8"
        self.sendbuf = torch.empty((max(shard.n_send, 1), 2), dtype=self.dtype, device="cuda")
        self._views = {}

    def step(self, nsteps, dt, model, stream):
        self.plan.run(nsteps, dt, model, stream.cuda_stream if stream is not None else None)

    def pack(self, stream, which=0):
        self.plan.pack_halo(self.sendbuf.data_ptr(), stream.cuda_stream if stream is not None else None, which)
        return self.sendbuf[:self.shard.n_send]

    def tile_classes(self):
        """(interior, boundary, ghost) tile counts of a fused plan, None for a sliced-ELL plan"""
        return self.plan.tile_classes() if self.plan.tile_info()["ksteps"] > 0 else None

    def launch(self, first, count, nsteps, dt, model, stream):
        self.plan.launch_tiles(first, count, nsteps, dt, model, stream.cuda_stream if stream is not None else None)

    def flip(self):
        self.plan.flip()

    def ghost_view(self, which=0):
        """tensor aliasing the ghost segment of the plan's current (which = 0) or next (1) state buffer"""
        ptr = self.plan.state_buffer(which)
        if ptr not in self._views:
            full = self.torch.as_tensor(_DevView(ptr, (self.shard.n_local, 2), self.typestr), device="cuda")
            self._views[ptr] = full[self.shard.n_own:]
        return self._views[ptr]

    def get_owned(self):
        n, c = self.plan.get_state()
        return n[:self.shard.n_own], c[:self.shard.n_own]


def exchange_halo(stepper, shard, stream=None, group=None, which=0, async_op=False, host_group=None):
    """one all-to-all-v of boundary state: each rank's send vertices -> the ghost segments of its
    neighbours' state buffers (which = 0 current, 1 next).  RCCL (backend nccl) moves device
    memory directly and, with async_op, returns a work handle: kernels launched afterwards on the
    compute stream run beside the exchange until handle.wait().  With gloo (CPU tests, several
    ranks sharing one GPU, or `host_group` = the gloo group ShardedRun falls back to when RCCL
    cannot be brought up) the buffers are staged through the host, synchronously.
    Stream order: the pack kernel and the tiles run on `stream` (None = the null stream), while
    torch.distributed orders a collective -- and a later handle.wait() -- against torch's CURRENT
    stream; so both are issued with `stream` made current (wait_exchange for the wait)."""
    import torch
    import torch.distributed as dist
    send = stepper.pack(stream, which) if which else stepper.pack(stream)
    recv = stepper.ghost_view(which) if which else stepper.ghost_view()
    in_splits, out_splits = [int(x) for x in shard.send_counts], [int(x) for x in shard.recv_counts]
    if host_group is not None:
        group = host_group
    if send.is_cuda and (host_group is not None or dist.get_backend(group) == "gloo"):
        torch.cuda.synchronize()
        send_h, recv_h = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(recv_h, send_h, out_splits, in_splits, group=group)
        recv.copy_(recv_h)
        torch.cuda.synchronize()
        return None
    if not send.is_cuda:
        return dist.all_to_all_single(recv, send.contiguous(), out_splits, in_splits, group=group, async_op=async_op)
    with torch.cuda.stream(_torch_stream(stream)):
        return dist.all_to_all_single(recv, send.contiguous(), out_splits, in_splits, group=group, async_op=async_op)


def _torch_stream(stream):
    """the torch stream object the library launches on: `stream`, or the null stream for None"""
    import torch
    return stream if stream is not None else torch.cuda.default_stream()


def wait_exchange(work, stream=None, on_device=True):
    """make `stream` (not whatever torch's current stream happens to be) wait for an asynchronous exchange"""
    if work is None:
        return
    if not on_device:
        work.wait()
        return
    import torch
    with torch.cuda.stream(_torch_stream(stream)):
        work.wait()


class ShardedRun:
    """the mesh split along the partition ranges over `world` ranks; k ghost rings let every rank
    take k steps between two halo exchanges (communication-avoiding explicit stepping)"""
    def __init__(self, mesh, nparts, rank, world, precision="f32", seed=1, krings=None, make_stepper=None,
                 ksteps=0, dt_rule="reference", control_group=None):
        self.ksteps = ksteps
        self.kernel_name = "tile_step_kernel" if ksteps > 0 else "sell_step_kernel"
        self.steps_per_launch = max(1, ksteps)
        # the partition is shared (deterministic: every rank computes the same one); everything after it is
        # rank-local: the shard, then the assembly of the rows of the shard's local vertices only
        self.part = _partition(mesh, nparts)
        om = self.part.ordered
        self.model = fa.default_model()
        self.precision = precision
        self.krings = krings or int(os.environ.get("FEM_HALO_RINGS", str(2 * ksteps if ksteps > 0 else 8)))   # two fused launches per exchange
        self.shard = fa.Shard(self.part, rank, world, self.krings)
        om.init_stats_shard(fa.PREC_F64, self.shard)
        self.dt = self._time_step(om, dt_rule, world, control_group)
        n0, c0 = fa.initial_condition(om.V, seed)                 # ordered numbering, identical on all ranks
        l2g = self.shard.loc2glob
        if make_stepper is None:
            self.stepper = HipShardStepper(om, self.shard, precision, self.model, n0[l2g], c0[l2g], part=self.part,
                                           ksteps=ksteps)
        else:
            self.stepper = make_stepper(om, self.shard, precision, self.model, n0[l2g], c0[l2g])
        self.plan = getattr(self.stepper, "plan", None)
        self.V = om.V
        self.budget = self.krings                                  # ghosts are fresh: k steps allowed
        self.pending, self.pending_exchange = None, False          # exchange in flight (overlapped schedule)
        self.pending_stream = None                                 # the stream its consumers run on
        tc = getattr(self.stepper, "tile_classes", None)
        self.classes = tc() if (tc is not None and os.environ.get("FEM_OVERLAP", "1") != "0") else None
        self.force_split = os.environ.get("FEM_OVERLAP", "1") == "split"   # tests: always take the not-yet-arrived path
        self.force_sync = os.environ.get("FEM_OVERLAP", "1") == "sync"     # tests: blocking exchange, so always the arrived path
        # one exchange up front (re-sends the ghosts' own initial values): brings up the RCCL
        # communicator and its buffers outside any timed region.  `control_group` (a gloo group over
        # the same ranks) makes that bring-up survivable: if RCCL raises here on any rank, all ranks
        # agree over gloo to stage every exchange through the host instead (slower, same results;
        # `host_group`/`fallback_reason` say so and bench.py reports it).
        import torch.distributed as dist
        self.host_group, self.fallback_reason = None, None
        if dist.is_available() and dist.is_initialized() and world > 1:
            err = None
            try:
                exchange_halo(self.stepper, self.shard, None)
                if control_group is not None:
                    import torch
                    torch.cuda.synchronize()
            except Exception as ex:                    # noqa: BLE001 (reported below or re-raised)
                if control_group is None:
                    raise
                err = ex
            if control_group is not None:
                import torch
                failed = torch.tensor([0 if err is None else 1])
                dist.all_reduce(failed, op=dist.ReduceOp.MAX, group=control_group)
                if int(failed.item()):
                    self.host_group = control_group
                    self.fallback_reason = ("%s: %s" % (type(err).__name__, str(err).strip().splitlines()[0][:160])
                                            if err is not None else "RCCL failed on another rank")
                    exchange_halo(self.stepper, self.shard, None, host_group=control_group)
        # what a launch advances for the job: the rows this rank owns (ghost rows are redundant work, not output)
        self.vertices_per_launch = self.shard.n_own
        self.seed = seed
        sb = 4 if precision == "f32" else 8
        plan = self.plan
        self.layout_bytes_per_launch = (plan.operator_bytes() + 4 * sb * self.shard.n_local) if plan else None

    def _time_step(self, om, rule, world, control_group=None):
        """_time_step on rank-local stats: maxEdgeLen is the whole mesh's on every rank; the "stable" rule needs the
        largest |wVtx| of the whole mesh = the maximum over the ranks' owned rows (one all-reduce at set-up)"""
        st = om.stats()
        if rule != "stable":
            return 0.1 * st["maxEdgeLen"] ** 2
        wmax = float(np.abs(st["wVtx"][self.shard.own_begin:self.shard.own_end]).max())
        import torch.distributed as dist
        if world > 1 and dist.is_available() and dist.is_initialized():
            import torch
            # over the gloo control group when there is one: RCCL may not be up yet (see the priming exchange)
            dev = "cuda" if control_group is None and dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor([wmax], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=control_group)
            wmax = float(t.item())
        return 0.2 / (max(1.0, self.model.D) * wmax)

    def run(self, nsteps, stream=None):
        if self.classes is not None:
            return self._run_overlapped(nsteps, stream)
        done = 0
        while done < nsteps:
            if self.budget == 0:
                exchange_halo(self.stepper, self.shard, stream, host_group=self.host_group)
                self.budget = self.krings
            chunk = min(self.budget, nsteps - done)
            self.stepper.step(chunk, self.dt, self.model, stream)
            self.budget -= chunk
            done += chunk

    def _run_overlapped(self, nsteps, stream):
        """fused plans: the tiles come as [interior | boundary | ghost].  The group of steps that
        uses up the ghost budget runs the boundary tiles first, packs and starts the exchange, and
        lets the interior tiles run beside it; the ghost tiles are skipped there (the exchange
        overwrites them).  The first group after an exchange runs the interior tiles (they read
        no ghost row) before it waits for the exchange.
        On the GPU the boundary tiles, the pack and the exchange go to a side stream
        and the interior tiles of the same group are launched on `stream` at once: the two launches
        share the chip (the few boundary workgroups are dispatched first, the interior ones fill the
        other slots) instead of a short boundary launch with its tail in front of the interior one."""
        nI, nB, nG = self.classes
        st, dt, model = self.stepper, self.dt, self.model
        side = self._side_stream(stream)
        done = 0
        while done < nsteps:
            s = min(self.ksteps, self.budget, nsteps - done)
            first, last = self.pending_exchange, self.budget == s
            merged = False
            if side is not None:
                self._ev_top.record(stream)                  # everything of the groups before this one
                if self._side_busy:
                    stream.wait_event(self._ev_side)         # ... including their boundary tiles
                    self._side_busy = False
            if first:
                # the exchange has had the previous group's interior launch to finish: if it has,
                # this group is ONE launch (no short boundary launch with its tail); if not, the
                # interior tiles go first and the wait comes after them
                arrived = self.pending is None or getattr(self.pending, "is_completed", lambda: False)()
                if not self.force_split and arrived:
                    self._wait_exchange(also=side if last else None)
                    merged = True
                else:
                    st.launch(0, nI, s, dt, model, stream)
                    self._wait_exchange(also=side if last else None)
            if last:
                bs = side if side is not None else stream
                if side is not None:
                    side.wait_event(self._ev_top)
                st.launch(nI, nB, s, dt, model, bs)
                self.pending = exchange_halo(st, self.shard, bs, which=1, async_op=not self.force_sync,
                                             host_group=self.host_group)
                self.pending_exchange, self.pending_stream = True, stream
                if side is not None:
                    self._ev_side.record(side)
                    self._side_busy = True
                if not first or merged:
                    st.launch(0, nI, s, dt, model, stream)
                self.budget = self.krings
            else:
                if first and not merged:
                    st.launch(nI, nB + nG, s, dt, model, stream)
                else:
                    st.launch(0, nI + nB + nG, s, dt, model, stream)
                self.budget -= s
            st.flip()
            done += s
        if side is not None and self._side_busy:             # leave `stream` ordered behind everything issued
            stream.wait_event(self._ev_side)
            self._side_busy = False

    def _side_stream(self, stream):
        """stream for the boundary tiles + pack + exchange of a group (None: CPU steppers, the null stream,
        FEM_BOUNDARY_STREAM=0).  Normal priority: see fem_amd/csrc/dist.cpp and DESIGN.md section 5 for why not high."""
        if stream is None or not isinstance(self.stepper, HipShardStepper) or os.environ.get("FEM_BOUNDARY_STREAM", "1") == "0":
            return None
        if getattr(self, "_side", None) is None:
            import torch
            self._side = torch.cuda.Stream()
            self._ev_top, self._ev_side = torch.cuda.Event(), torch.cuda.Event()
            self._side_busy = False
        return self._side

    def _wait_exchange(self, also=None):
        """`also`: a second stream whose next work reads the ghost rows too (the boundary tiles' side stream)"""
        if self.pending is not None:
            on_device = isinstance(self.stepper, HipShardStepper)
            wait_exchange(self.pending, self.pending_stream, on_device=on_device)
            if also is not None:
                wait_exchange(self.pending, also, on_device=on_device)
        self.pending, self.pending_exchange = None, False

    def get_owned(self):
        self._wait_exchange()
        return self.stepper.get_owned()

    def reset(self):
        """back to the seeded initial condition on owned and ghost rows alike, ghost budget full again"""
        if self.plan is None:
            raise fa.FemError("ShardedRun.reset: only for plans on the device")
        self._wait_exchange()
        n0, c0 = fa.initial_condition(self.V, self.seed)
        l2g = self.shard.loc2glob
        self.plan.set_state(n0[l2g], c0[l2g])              # synchronises the device first
        self.budget = self.krings

    def traffic_key(self):
        ti = self.plan.tile_info() if self.plan is not None else None
        return "%s:%s:shard %d of %d" % (self.kernel_name, self.precision, self.shard.rank, self.shard.nranks) + (
            ":k%d:%dx%d" % (ti["ksteps"], ti["threads"], ti["rows_per_thread"]) if ti else "")

    def measured_traffic(self):
        return None, None
