"""How a 7-ring fused launch splits into loading and stepping: the bench plan (16.4M sphere, k = 7 layout) launched
with s = 1..7 steps per launch; T(s) = a + b s.  usage (GPU box): python tools/step_slope_probe.py [workload]"""
import sys, time, json
import torch
import fem_amd as fa
from fem_amd.dist import SingleRun

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
pts, tri = fa.make_icosphere(nu, 0.1)
mesh = fa.Mesh(pts, tri)
run = SingleRun(mesh, max(1, mesh.V // 975), "f32", ksteps=7)
plan = run.plan
stream = torch.cuda.current_stream().cuda_stream
res = {}
for s in [7, 1, 2, 3, 4, 5, 6, 7]:
    for _ in range(3):
        plan.run(s, run.dt, run.model, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 60
    e0.record()
    for _ in range(n):
        plan.run(s, run.dt, run.model, stream)
    e1.record()
    torch.cuda.synchronize()
    res[s] = e0.elapsed_time(e1) / n * 1e3
    print("steps per launch %d: %.1f us per launch" % (s, res[s]), flush=True)
b = (res[7] - res[1]) / 6
print(json.dumps({"us_per_launch": res, "slope_us_per_step": b, "intercept_us": res[1] - b}))
