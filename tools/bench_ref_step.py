#!/usr/bin/env python3
"""Rate of the drop-in `step` path (reference layout, ref_step_kernel): tools/bench_ref_step.py [nu=320] [verts/part=256] [steps=200]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fem_amd as fa

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 320
vpp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
pts, tri = fa.make_icosphere(nu, 0.1)
mesh = fa.Mesh(pts, tri)
part = mesh.partition(max(1, mesh.V // vpp))
om = part.ordered
om.init_stats(fa.PREC_MIXED)
dt = 0.1 * om.stats()["maxEdgeLen"] ** 2
lay = fa.RefLayout(part)
dev = fa.RefDevice(lay)
n0, c0 = fa.initial_condition(om.V, 1)
dev.set_state(n0, c0)
dev.run(4, dt)
torch.cuda.synchronize()
t0 = time.perf_counter()
dev.run(steps, dt)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / steps
print(json.dumps({"V": om.V, "partitions": part.n, "maxV": lay.maxV, "maxF": lay.maxF, "ms_per_step": el * 1e3,
                  "DOF_it_per_s": 2 * om.V / el}))
