"""`python bench.py --gpus N` from a bare shell (no torchrun): the parent starts N fresh rank processes before any
GPU call, relays rank 0's line and exits non-zero if any rank does (the reference's harness is one
process that times itself, src/main.cpp:547-575).  CPU only: the ranks here are gloo workers, not bench.py itself
(bench.py's ranks need the GPU; tests/test_sharded.py runs that form on the GPU box)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "launch_worker.py")
DRIVER = ("import sys; sys.path.insert(0, %r); import bench; "
          "sys.exit(bench.launch_ranks(int(sys.argv[1]), [sys.executable, %r] + sys.argv[2:], grace=3.0))" % (ROOT, WORKER))


def run(world, *extra, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FEM_HOST_THREADS")}
    return subprocess.run([sys.executable, "-c", DRIVER, str(world)] + list(extra), capture_output=True, text=True,
                          timeout=timeout, env=env)


def test_launcher_starts_ranks_and_relays_rank0():
    res = run(3)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["sum"] == 6.0 and out["world"] == 3 and out["master"] == "127.0.0.1" and out["local_rank"] == 0
    assert 2 <= int(out["host_threads"]) <= 16          # the ranks share the host's cores
    assert "noise from rank 0" in res.stdout and "noise from rank 1" not in res.stdout
    assert "noise from rank 1" in res.stderr and "noise from rank 2" in res.stderr


def test_launcher_fails_when_a_rank_fails_and_stops_the_others():
    t0 = time.monotonic()
    res = run(3, "--fail-rank", "1", "--hang")
    assert res.returncode == 3, (res.returncode, res.stderr[-2000:])
    assert time.monotonic() - t0 < 60                   # the hanging ranks were stopped, not waited for
    assert "rank 1 exited with status 3" in res.stderr


def test_launcher_module_does_not_touch_the_gpu_or_torch():
    """the parent must stay clear of the GPU: importing bench (and launching) pulls in neither torch nor the library"""
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "assert 'torch' not in sys.modules and 'fem_amd' not in sys.modules; print('clean')" % ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "clean" in res.stdout, res.stderr[-1000:]


def test_bench_refuses_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=60, env=env)
    assert res.returncode != 0 and "WORLD_SIZE=3" in res.stderr
