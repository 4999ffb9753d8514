"""worker of the multi-process tests: one rank of a ShardedRun (launched by torch.distributed.run)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

ap = argparse.ArgumentParser()
ap.add_argument("--backend", choices=["oracle", "hip"], required=True)
ap.add_argument("--sphere", type=int, default=10)
ap.add_argument("--nparts", type=int, default=8)
ap.add_argument("--krings", type=int, default=2)
ap.add_argument("--steps", type=int, default=7)
ap.add_argument("--precision", default="f64")
ap.add_argument("--fuse", type=int, default=0)
ap.add_argument("--pg", choices=["gloo", "nccl"], default="gloo")
ap.add_argument("--out", required=True)
a = ap.parse_args()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import fem_amd as fa  # noqa: E402
from fem_amd.dist import ShardedRun  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if a.pg == "nccl":
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dist.init_process_group(a.pg)
rank, world = dist.get_rank(), dist.get_world_size()
make = None
stream = None
if a.backend == "oracle":
    if a.fuse > 0:                                 # drives the overlapped tile schedule of ShardedRun on the CPU
        from shard_oracle import OracleTileStepper

        def make(*args):
            return OracleTileStepper(*args, ksteps=a.fuse)
    else:
        from shard_oracle import OracleShardStepper as make
else:
    if a.pg == "gloo":
        torch.cuda.set_device(0)                  # all ranks share the one GPU of the test box
    stream = torch.cuda.current_stream()
pts, tri = fa.make_icosphere(a.sphere, 0.1)
run = ShardedRun(fa.Mesh(pts, tri), a.nparts, rank, world, a.precision, seed=3, krings=a.krings, make_stepper=make,
                 ksteps=a.fuse)
# split the run so that exchanges fall both inside and between run() calls
run.run(a.steps // 2, stream)
run.run(a.steps - a.steps // 2, stream)
n, c = run.get_owned()
np.savez(a.out + ".%d.npz" % rank, n=n, c=c, own_begin=run.shard.own_begin, own_end=run.shard.own_end, dt=run.dt,
         classes=np.array(run.classes if run.classes is not None else (-1, -1, -1)),
         launches=np.array(getattr(run.stepper, "launch_log", []) or [(-1, -1, -1)]))
if a.pg == "nccl":
    dist.barrier(device_ids=[torch.cuda.current_device()])
else:
    dist.barrier()
dist.destroy_process_group()
