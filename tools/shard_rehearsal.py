#!/usr/bin/env python3
"""One rank of an N-rank sharded run on ONE GPU, exchange stubbed out (pack kernel only): measures
what the sharding costs on the device -- ghost rows, split launches, tails -- against 1/N of the
whole-mesh time.  The all-to-all itself (latency over xGMI) is not in this number; the overlapped
schedule hides it behind the interior tiles.
usage: tools/shard_rehearsal.py [workload=sphere16m] [world=8] [rank=0] [krings=8] [steps=400]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fem_amd as fa  # noqa: E402
from fem_amd import dist as fdist  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "sphere16m"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
krings = int(sys.argv[4]) if len(sys.argv) > 4 else 10
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 400
nu = {"sphere16m": 1280, "sphere4m": 640, "sphere1m": 320, "scan8m": 900, "scan500k": 224}[wl]
fuse = int(os.environ.get("FEM_FUSE_STEPS", "7"))


class _Done:
    """FEM_REHEARSE_PENDING=1: pretend the exchange is still in flight when the next group starts"""
    def wait(self):
        pass

    def is_completed(self):
        return os.environ.get("FEM_REHEARSE_PENDING", "0") != "1"


def stub_exchange(stepper, shard, stream=None, group=None, which=0, async_op=False, host_group=None):
    stepper.pack(stream, which)          # the device-side work of an exchange; no bytes leave the GPU
    return _Done()


fdist.exchange_halo = stub_exchange
pts, tri = fa.make_scanned_like(nu, 0.1, seed=3) if wl.startswith("scan") else fa.make_icosphere(nu, 0.1)
mesh = fa.Mesh(pts, tri)
V = mesh.V
run = fdist.ShardedRun(mesh, max(world, V // int(os.environ.get("FEM_TILE_N0", {5: "1200", 6: "1050", 7: "975"}.get(fuse, "975")))), rank, world, "f32", krings=krings,
                       ksteps=fuse, dt_rule="stable" if wl.startswith("scan") else "reference")
del pts, tri, mesh
stream = torch.cuda.current_stream()
run.run(2 * krings, stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
run.run(steps, stream)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
sh = run.shard
print(json.dumps({"workload": wl, "world": world, "rank": rank, "krings": krings, "fuse": fuse, "classes": run.classes,
                  "n_own": sh.n_own, "n_ghost": sh.n_local - sh.n_own, "n_send": sh.n_send,
                  "ms_per_step": dt / steps * 1e3, "rank_vertex_steps_per_s": sh.n_own * steps / dt,
                  "job_DOF_it_per_s_if_all_ranks_alike": 2.0 * V * steps / dt}))
