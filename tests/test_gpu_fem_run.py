"""fem_run: the command-line driver over the C-ABI (fem_amd/csrc/fem_run.cpp) that follows the
reference's main / GPUrun / CPUrun protocol (main.cpp:116-154, 344-617, 619-676).  Run here on
BASELINE config 1 as the reference would be: the nu=32 sphere from an .off file, 52 partitions."""
import os
import re
import subprocess

import numpy as np
import pytest

import fem_amd as fa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FEM_RUN = os.path.join(ROOT, "fem_amd", "lib", "fem_run")

ROUND = re.compile(r"Round (\d): Sum / Mean error for n: (\S+) / (\S+) and for c: (\S+) / (\S+)\. "
                   r"Number of errors in total: (\d+)\((\d+)\)")


def run(*args, timeout=120):
    return subprocess.run([FEM_RUN, *args], capture_output=True, text=True, timeout=timeout)


def test_usage_without_arguments():
    assert os.access(FEM_RUN, os.X_OK), "fem_run not built (python __graft_entry__.py)"
    res = run()
    assert res.returncode == 2 and "usage: fem_run" in res.stderr


def test_missing_file_is_an_error_not_a_crash(tmp_path):
    res = run(str(tmp_path / "nothing.off"))
    assert res.returncode == 1 and "nothing.off" in res.stderr


@pytest.fixture(scope="module")
def sphere_off(tmp_path_factory):
    pts, tri = fa.make_icosphere(32, 0.1)
    path = str(tmp_path_factory.mktemp("fem_run") / "sphere32.off")
    fa.off_write(path, pts, tri)
    return path, len(pts), len(tri)


@pytest.mark.gpu
def test_reference_protocol_on_config1(sphere_off, tmp_path):
    path, V, F = sphere_off
    out = str(tmp_path / "pattern.off")
    res = run(path, "-v", "--iterations", "100", "--out", out)
    assert res.returncode == 0, res.stdout + res.stderr
    rounds = {int(m.group(1)): m for m in ROUND.finditer(res.stdout)}
    assert set(rounds) == {1, 2}, res.stdout
    for m in rounds.values():
        assert int(m.group(7)) == V
        assert float(m.group(3)) < 1e-10 and float(m.group(5)) < 1e-10      # mean |step() - host operators| per vertex
    m = re.search(r"Time needed for 100 iterations: (\S+) ms with (\d+) threads each and 52 partitions", res.stdout)
    assert m and float(m.group(1)) > 0, res.stdout
    best = re.search(r"Best time with 52 partitions is (\S+)!", res.stdout)
    assert best and float(best.group(1)) > 0
    pts, tri, rgb = fa.off_read(out)
    assert len(pts) == V and len(tri) == F and rgb is not None
    s = rgb[:, 0].astype(int) + rgb[:, 1].astype(int)
    assert np.all((s == 254) | (s == 255)) and np.array_equal(rgb[:, 1], rgb[:, 2])   # the colour law of formatKernel


@pytest.mark.gpu
def test_debug_flag_checks_only(sphere_off):
    res = run(sphere_off[0], "-d")
    assert res.returncode == 0 and "Round 2" in res.stdout
    assert "Best time" not in res.stdout and "Time:" not in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [(), ("--fuse", "0")])
def test_native_plan_same_protocol(sphere_off, extra):
    res = run(sphere_off[0], "--native", "-v", "--iterations", "50", "--rows", "400", *extra)
    assert res.returncode == 0, res.stdout + res.stderr
    m = ROUND.search(res.stdout)
    assert m and int(m.group(1)) == 2
    assert float(m.group(3)) < 2e-7 and float(m.group(5)) < 2e-7             # fp32 state against the fp64 host steps
    assert re.search(r"Best time with 25 partitions is (\S+)!", res.stdout)


@pytest.mark.gpu
def test_partition_limit_is_reported_like_the_reference(sphere_off):
    res = run(sphere_off[0], "-n", "4")                                      # 2560 vertices per partition > 1024
    assert res.returncode == 0
    assert "Too many nodes or elements per Block" in res.stdout
    assert "Best time with -1 partitions" in res.stdout


@pytest.mark.gpu
def test_ply_input_and_partition_range(tmp_path):
    """binary PLY in (the format scans ship in), two partition counts tried, best one reported"""
    pts, tri = fa.make_icosphere(8, 0.1)
    path = str(tmp_path / "s.ply")
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                 "property float z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n"
                 % (len(pts), len(tri))).encode())
        f.write(pts.astype("<f4").tobytes())
        rec = np.empty(len(tri), dtype=[("n", "u1"), ("v", "<i4", 3)])
        rec["n"], rec["v"] = 3, tri
        f.write(rec.tobytes())
    res = run(path, "-n", "3:4", "--iterations", "20")
    assert res.returncode == 0, res.stdout + res.stderr
    lines = re.findall(r"Errors: (\d+), Partitions: (\d+), Threads: (\d+), Time: (\S+)", res.stdout)
    assert [int(l[1]) for l in lines] == [3, 4]
    best = re.search(r"Best time with (\d+) partitions is (\S+)!", res.stdout)
    times = {int(l[1]): float(l[3]) for l in lines}
    assert best and int(best.group(1)) == min(times, key=times.get)


@pytest.mark.gpu
def test_native_plan_numbers_wide_rows_first(tmp_path):
    """an irregular mesh (valences up to 16) through `fem_run --native`: the driver numbers the rows of more than 6 entries
    first (fem_mesh_wide_rows_first, as fem_amd/dist.py:_partition does), says so with -v, and the fused plan on the
    renumbered mesh passes the program's own check against its fp64 host steps; a regular sphere is left as it is"""
    pts, tri = fa.make_scanned_like(24, 0.1, seed=3, high_valence_fraction=0.01)
    path = str(tmp_path / "scan.off")
    fa.off_write(path, pts, tri)
    res = run(path, "--native", "-v", "--iterations", "20", "--rows", "400", "--fuse", "4", "--dt-scale", "0.2")
    assert res.returncode == 0, res.stdout + res.stderr
    m = re.search(r"(\d+) rows of more than 6 entries: numbered first", res.stdout)
    val = np.bincount(tri.ravel().astype(np.int64), minlength=len(pts))
    assert m and int(m.group(1)) == int((val > 6).sum())
    r = ROUND.search(res.stdout)
    assert r and float(r.group(3)) < 2e-7 and float(r.group(5)) < 2e-7
    sp, st = fa.make_icosphere(8, 0.1)
    path2 = str(tmp_path / "sphere.off")
    fa.off_write(path2, sp, st)
    res2 = run(path2, "--native", "-v", "--iterations", "10", "--rows", "200")
    assert res2.returncode == 0 and "0 rows of more than 6 entries" in res2.stdout and "numbered first" not in res2.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("flag,bar", [("--f64", 1e-10), ("--mixed", 2e-10)])
def test_native_plan_in_fp64_and_in_the_reference_mix(sphere_off, flag, bar):
    """`fem_run --f64` / `--mixed` without --ranks: the native single-GPU fused plan in fp64, and in the reference's own
    precision mix (FEM_PLAN_MIXED_RECORDS), each passing the program's check against its host steps at its own level.  The
    host steps are the reference's CPUrun, whose gradient dot product is fp32 (main.cpp:647-665), so even the fp64 plan
    differs from them: mean error per vertex of n after two steps 2.6e-11 (fp64), 5.4e-11 (mixed), 6.4e-9 (fp32 plan)"""
    res = run(sphere_off[0], flag, "-v", "--iterations", "20", "--rows", "400", "--fuse", "4")
    assert res.returncode == 0, res.stdout + res.stderr
    m = ROUND.search(res.stdout)
    assert m and float(m.group(3)) < bar and float(m.group(5)) < bar, res.stdout
