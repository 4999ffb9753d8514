#!/usr/bin/env python3
"""What the fused kernel pays for a partition: edge cut, cut-face fraction (the reference's bucket n,
partitioning_common.cpp:78-100), part-size spread and the ring redundancy m[k-1]/m[0] of the k-step tiles (rows within
k-1 edges of a part / rows of the part; sampled over parts), for the three R0 stages: recursive coordinate bisection,
graph bisection, multilevel k-way on the dual graph.  CPU only.
usage: tools/partition_quality.py [sphere|scan|roll] [nu] [rows per part] [k]
(roll: a 1200 x 400 sheet rolled into a spiral with layers 0.35 apart -- coordinates that mislead)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import fem_amd as fa  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "sphere"
nu = int(sys.argv[2]) if len(sys.argv) > 2 else 320
n0 = int(sys.argv[3]) if len(sys.argv) > 3 else 975
k = int(sys.argv[4]) if len(sys.argv) > 4 else 7
def rolled_sheet(nx=1200, ny=400, h=0.1, gap=0.35):
    u = np.arange(nx) * h
    b = gap / (2 * np.pi)
    theta = np.sqrt(2 * (u + 5.0) / b)
    r = b * theta
    x, z = r * np.cos(theta), r * np.sin(theta)
    p = np.stack([np.repeat(x[:, None], ny, 1).ravel(), np.repeat((np.arange(ny) * h)[None, :], nx, 0).ravel(),
                  np.repeat(z[:, None], ny, 1).ravel()], 1)
    idx = np.arange(nx * ny).reshape(nx, ny)
    a, b_, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    return p, np.concatenate([np.stack([a, b_, c], 1), np.stack([a, c, d], 1)]).astype(np.uint32)


pts, tri = rolled_sheet() if kind == "roll" else (fa.make_scanned_like(nu, 0.1, seed=3) if kind == "scan" else fa.make_icosphere(nu, 0.1))
mesh = fa.Mesh(pts, tri)
n = mesh.V // n0
for method, env in (("rcb", None), ("graph", None), ("auto", None), ("multilevel", None), ("multilevel", "graph")):
    if env:
        os.environ["FEM_ML_INITIAL"] = env
    else:
        os.environ.pop("FEM_ML_INITIAL", None)
    t0 = time.perf_counter()
    npart = mesh.partition_vertices(n, method)
    secs = time.perf_counter() - t0
    part = mesh.partition(n, npart=npart)
    om = part.ordered
    topo = om.topology()
    A = sp.csr_matrix((np.ones(len(topo["col"]), np.int8), topo["col"].astype(np.int64), topo["rowptr"].astype(np.int64)),
                      shape=(om.V, om.V))
    sizes = np.diff(part.node_parts.astype(np.int64))
    fs = np.diff(part.element_parts.astype(np.int64))
    rng = np.random.default_rng(1)
    sample = rng.choice(n, size=min(n, 200), replace=False)
    red = []
    for p in sample:
        x = np.zeros(om.V, np.int8)
        x[part.node_parts[p]:part.node_parts[p + 1]] = 1
        for _ in range(k - 1):
            x = ((A @ x) + x > 0).astype(np.int8)
        red.append(x.sum() / sizes[p])
    print(json.dumps({"mesh": "%s nu=%d V=%d" % (kind, nu, om.V), "method": method + ("+graph-growing initial" if env else "") + (" -> " + mesh.last_partitioner if method == "auto" else ""),
                      "parts": n, "seconds": round(secs, 2), "vertices_min_mean_max": [int(sizes.min()), float(sizes.mean()), int(sizes.max())],
                      "edge_cut": int(part.cut), "cut_face_fraction": float(fs[-1] / om.F),
                      "ring_redundancy_k%d_mean" % k: float(np.mean(red)), "ring_redundancy_max": float(np.max(red))}))
