"""Run drivers over the C-ABI: a whole mesh on one GPU (SingleRun) and a mesh sharded along
the partition ranges over the ranks of one node (ShardedRun, halo exchange through
torch.distributed).  Plumbing only: all arithmetic happens in libfem_hip.so."""
import json
import os

import numpy as np

import fem_amd as fa

_PROFILES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


class SingleRun:
    """partition -> reorder -> assemble (host, once) -> native plan; seeded IC of main.cpp:178-186"""
    kernel_name = "sell_step_kernel"

    def __init__(self, mesh, nparts, precision="f32", seed=1):
        self.part = mesh.partition(nparts)
        om = self.part.ordered
        om.init_stats(fa.PREC_F64)
        self.dt = 0.1 * om.stats()["maxEdgeLen"] ** 2          # main.cpp:188
        self.model = fa.default_model()
        self.precision = precision
        self.plan = fa.Plan(om, fa.PREC_F32 if precision == "f32" else fa.PREC_F64)
        n0, c0 = fa.initial_condition(om.V, seed)
        self.plan.set_state(n0, c0)
        self.V = om.V
        self.vertices_per_launch = om.V
        sb = 4 if precision == "f32" else 8
        self.layout_bytes_per_launch = self.plan.operator_bytes() + 4 * sb * om.V   # operator + state r/w

    def run(self, nsteps, stream=None):
        self.plan.run(nsteps, self.dt, self.model, stream.cuda_stream if stream is not None else None)

    def measured_traffic(self):
        """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (None if no
        pass was recorded for this mesh size)"""
        path = os.path.join(_PROFILES, "traffic.json")
        if not os.path.exists(path):
            return None
        with open(path) as f:
            rec = json.load(f)
        return rec.get("%s:%s:%d" % (self.kernel_name, self.precision, self.V))
