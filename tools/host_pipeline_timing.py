"""Host side of the path, stage by stage (no GPU): write .off / read .off / binary .ply, topology, partition + reorder
(R0-R3), assembly (initMeshStats), reference-layout pack.  usage: python tools/host_pipeline_timing.py [nu=640] [tmpdir=/tmp]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fem_amd as fa  # noqa: E402

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 640
tmp = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
out = {"nu": nu, "host_threads": int(os.environ.get("FEM_HOST_THREADS", "0")) or min(16, os.cpu_count() or 1)}


def timed(name, fn):
    t0 = time.perf_counter()
    r = fn()
    out[name + "_s"] = round(time.perf_counter() - t0, 3)
    return r


pts, tri = timed("make_icosphere", lambda: fa.make_icosphere(nu, 0.1))
out["V"], out["F"] = len(pts), len(tri)
off, ply = os.path.join(tmp, "timing.off"), os.path.join(tmp, "timing.ply")
timed("off_write", lambda: fa.off_write(off, pts, tri))
out["off_MB"] = round(os.path.getsize(off) / 1e6, 1)
p2, t2, _ = timed("off_read", lambda: fa.off_read(off))
if hasattr(fa, "ply_write"):
    timed("ply_write_binary", lambda: fa.ply_write(ply, pts, tri))
    timed("ply_read_binary", lambda: fa.read_mesh(ply))
mesh = timed("topology", lambda: fa.Mesh(p2, t2))
part = timed("partition_rcb_and_reorder", lambda: mesh.partition(max(1, mesh.V // 975)))
timed("assembly_initMeshStats_mixed", lambda: part.ordered.init_stats(fa.PREC_MIXED))
for f in (off, ply):
    if os.path.exists(f):
        os.remove(f)
print(json.dumps(out))
