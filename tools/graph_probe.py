"""Does capturing the launch loop in a HIP graph pay on launch-bound meshes?  fem_plan_run of N steps, eager against a
captured graph replayed (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the stream the library launches on).
usage (GPU box): PYTHONPATH=. python tools/graph_probe.py [nu=32] [steps=600]"""
import sys, time, json
import torch
import fem_amd as fa
from fem_amd.dist import SingleRun

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
pts, tri = fa.make_icosphere(nu, 0.1)
mesh = fa.Mesh(pts, tri)
k = 6
run = SingleRun(mesh, max(1, mesh.V // 760), "f32", ksteps=k)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run.plan.run(steps, run.dt, run.model, s.cuda_stream)
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run.plan.run(steps, run.dt, run.model, s.cuda_stream)
    s.synchronize()
    eager = (time.perf_counter() - t0) / 10
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        run.plan.run(steps, run.dt, run.model, s.cuda_stream)
    g.replay(); s.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    s.synchronize()
    graph = (time.perf_counter() - t0) / 10
print(json.dumps({"V": mesh.V, "steps": steps, "launches": -(-steps // k), "eager_us_per_step": eager / steps * 1e6,
                  "graph_us_per_step": graph / steps * 1e6, "eager_G": 2 * mesh.V * steps / eager / 1e9, "graph_G": 2 * mesh.V * steps / graph / 1e9}))
