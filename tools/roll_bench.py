#!/usr/bin/env python3
"""The rolled sheet (coordinates that mislead) through bench.py with the coordinate bisection and with the default `auto`
stage R0 (which switches to graph bisection there): what the partition costs the fused kernel, end to end, verified.
usage (GPU box): python tools/roll_bench.py [nx=2400] [ny=800]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fem_amd as fa

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 800
h, gap = 0.1, 0.35
u = np.arange(nx) * h
b = gap / (2 * np.pi)
theta = np.sqrt(2 * (u + 5.0) / b)
r = b * theta
x, z = r * np.cos(theta), r * np.sin(theta)
pts = np.stack([np.repeat(x[:, None], ny, 1).ravel(), np.repeat((np.arange(ny) * h)[None, :], nx, 0).ravel(),
                np.repeat(z[:, None], ny, 1).ravel()], 1)
idx = np.arange(nx * ny).reshape(nx, ny)
a, b_, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
tri = np.concatenate([np.stack([a, b_, c], 1), np.stack([a, c, d], 1)]).astype(np.uint32)
path = "/tmp/rolled_sheet.off"
fa.off_write(path, pts, tri)
for method in ("rcb", "auto"):
    env = dict(os.environ, FEM_PARTITIONER=method, FEM_FUSE_STEPS="6", FEM_TILE_N0="760")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mesh", path, "--steps", "300", "--warmup", "6",
                          "--no-cpu-baseline", "--sustained-steps", "0"], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if res.returncode != 0 or not lines:
        print(method, "FAILED", res.stderr[-1500:])
        sys.exit(1)
    o = json.loads(lines[0])
    print(json.dumps({"mesh": "rolled sheet %d x %d" % (nx, ny), "FEM_PARTITIONER": method, "G_DOF_it_per_s": o["value"] / 1e9,
                      "ms_per_step": o["ms_per_step"], "tiles": o["config"]["tiles"]["ntiles"], "rows_loaded": o["config"]["tiles"]["rows_loaded"],
                      "cut_faces_fraction": o["config"]["mesh"]["cut_faces_fraction"], "verify_ok": o["verify"]["ok"],
                      "verify_dev_n": o["verify"]["rel_l2_n_deviation"]}))
