"""The reference-named C++ host interfaces (include/fem_host.hpp: SimpleTriMesh, MeshStats, partition,
initMeshStats, compute*CPU, deleteMeshStats) called from a C++ translation unit of their own with the reference's
argument orders (partitioning_common.h:3-9, mesh_function.h:6-125), compiled by plain g++ and linked against
libfem_hip.so; its one CPUrun step is compared with the oracle's."""
import os
import subprocess

import numpy as np

import fem_amd as fa
import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_caller_of_the_reference_named_interfaces(tmp_path):
    exe = str(tmp_path / "host_interface")
    libdir = os.path.dirname(fa.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_interface.cpp"), "-o", exe, "-L", libdir, "-lfem_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    nu, n = 8, 4
    res = subprocess.run([exe, str(nu), str(n)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = res.stdout.splitlines()
    head = lines[0].split()
    V, F = int(head[1]), int(head[3])
    parts = [int(x) for x in lines[0].split("parts")[1].split("|")[0].split()]
    eparts = [int(x) for x in lines[0].split("elements")[1].split()]
    state = np.array([[float(x) for x in ln.split()] for ln in lines[2:]])
    # the same through the flat C-ABI + the oracle (mixed precision = the reference's Vec3f/double mix)
    pts, tri = fa.make_icosphere(nu, 0.1)
    part = fa.Mesh(pts, tri).partition(n)
    assert (V, F) == (part.ordered.V, part.ordered.F)
    assert parts == part.node_parts.tolist() and eparts == part.element_parts.tolist()
    st = ol.Stats(ol.Topology(V, part.ordered.tri), part.ordered.points, ol.MIXED)
    assert float(lines[1].split()[1]) == st.maxEdgeLen
    n0, c0 = ol.initial_condition(V, 1)
    n1, c1 = st.step(n0, c0, 0.1 * st.maxEdgeLen ** 2, 1, fa.default_model().as_array())
    assert state.shape == (V, 2)
    assert np.abs(state[:, 0] - n1).max() <= 1e-12 and np.abs(state[:, 1] - c1).max() <= 1e-12
