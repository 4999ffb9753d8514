#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run from the repo root, CPU only).

  partition_golden.npz : vertex -> part of the product's recursive-coordinate-bisection split (regression pin of the
                         stage METIS occupied in the reference; there is no reference output
                         to compare with: METIS is absent and its version unpinned).
  step_golden.npz      : (n, c) after {1, 10, 100} steps from the seeded IC, computed by the
                         all-fp64 oracle (oracle/fem_oracle.c) on the nu=32 sphere reordered by
                         the golden partition (P = 52) and on the 451-vertex strip (P = 3); the
                         oracle itself is pinned by tests/test_oracle_kat.py and
                         tests/test_oracle_scipy.py.  Stored as float64.
ResultantPattern.off is a data file copied from the reference (its only output artefact).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fem_amd as fa  # noqa: E402
import oracle_lib as ol  # noqa: E402
from helpers import sphere_case, strip_case  # noqa: E402

out = os.path.dirname(os.path.abspath(__file__))

g = {}
for name, mk, n in (("ico9", lambda: fa.make_icosphere(9, 0.1), 7), ("strip", lambda: fa.make_strip(11, 41, 0.1), 3)):
    pts, tri = mk()
    g["%s_n%d" % (name, n)] = fa.Mesh(pts, tri).partition_vertices(n).astype(np.int16)
np.savez_compressed(os.path.join(out, "partition_golden.npz"), **g)

s = {}
for name, case in (("sphere32_p52", sphere_case(32, 52)), ("strip_p3", strip_case(3))):
    st = case.ostats(ol.F64)
    s[name + "_perm_v"] = case.part.perm_v.astype(np.int32)
    s[name + "_dt"] = np.array(case.dt)
    n, c, done = case.n0, case.c0, 0
    for steps in (1, 10, 100):
        n, c = st.step(n, c, case.dt, steps - done, case.model.as_array())
        done = steps
        s["%s_n_%d" % (name, steps)], s["%s_c_%d" % (name, steps)] = n, c
np.savez_compressed(os.path.join(out, "step_golden.npz"), **s)
for f in ("partition_golden.npz", "step_golden.npz"):
    print(f, os.path.getsize(os.path.join(out, f)))
