#!/usr/bin/env python3
"""Host cost of one exchange period of ShardedRun's overlapped schedule WITH torch.distributed's collective in it
(backend nccl = RCCL at world size 1: the all-to-all moves nothing, its host path is the full one).  A tiny shard, so
the GPU is never the bottleneck: wall time per period = what the host needs to issue it.  At 8 ranks a period of the
16.4M sphere is ~220 us of GPU time (14 steps); the host must stay below that to keep the GPU fed.
usage: tools/host_period_probe.py [periods=2000]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
import fem_amd as fa
from fem_amd import dist as fdist

periods = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
pts, tri = fa.make_icosphere(24, 0.1)
mesh = fa.Mesh(pts, tri)
run = fdist.ShardedRun(mesh, 12, 0, 1, "f32", ksteps=7)
stream = torch.cuda.current_stream()
k = run.krings
run.run(20 * k, stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
run.run(periods * k, stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"periods": periods, "steps_per_period": k, "classes": run.classes,
                  "host_us_per_period": (t1 - t0) / periods * 1e6, "wall_us_per_period": (t2 - t0) / periods * 1e6}))
dist.destroy_process_group()
