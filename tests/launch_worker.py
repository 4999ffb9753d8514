"""worker of tests/test_bench_launcher.py: one rank started by bench.launch_ranks (no GPU, gloo)"""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--fail-rank", type=int, default=-1)
ap.add_argument("--hang", action="store_true", help="the ranks that do not fail wait for ever (the launcher must stop them)")
a = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if rank == a.fail_rank:
    print("rank %d fails on purpose" % rank, file=sys.stderr)
    sys.exit(3)
if a.hang:
    time.sleep(600)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dist.init_process_group("gloo")
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
print("noise from rank %d" % rank)                 # only rank 0's stdout reaches the launcher's stdout
if rank == 0:
    print(json.dumps({"sum": float(t), "world": world, "local_rank": int(os.environ["LOCAL_RANK"]),
                      "master": os.environ["MASTER_ADDR"], "host_threads": os.environ.get("FEM_HOST_THREADS")}), flush=True)
dist.barrier()
dist.destroy_process_group()
