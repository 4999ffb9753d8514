"""Accuracy of the 12-byte operator records (FEM_TILE_PACKED) against the fp32 records and the f64 oracle:
rel-L2 of n after a number of steps, from the linear regime into the developed pattern.
usage (GPU box): PYTHONPATH=. python tests/report_packed_accuracy.py [nu | scan:nu] [steps ...]
(scan:nu = the irregular "scanned-like" mesh of config 4 with its stable time step and the 1536-row tile)"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import fem_amd as fa
import oracle_lib as ol

scan = len(sys.argv) > 1 and sys.argv[1].startswith("scan:")
nu = int(sys.argv[1].split(":")[-1]) if len(sys.argv) > 1 else 160
checkpoints = [int(a) for a in sys.argv[2:]] or [25, 100, 400, 1600]
pts, tri = fa.make_scanned_like(nu, 0.1, seed=3) if scan else fa.make_icosphere(nu, 0.1)
mesh = fa.Mesh(pts, tri)
part = mesh.partition(max(1, mesh.V // (820 if scan else 975)))
om = part.ordered
om.init_stats(fa.PREC_MIXED)
dt = 0.1 * om.stats()["maxEdgeLen"] ** 2
model = fa.default_model()
if scan:
    from fem_amd.dist import _time_step
    dt = _time_step(om, model, "stable")
n0, c0 = fa.initial_condition(om.V, 1)
st = ol.Stats(ol.Topology(om.V, om.tri), om.points, ol.F64)
ol.set_threads(16)

def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))

plans = {}
for name, env in (("packed", "1"), ("fp32", "0")):
    os.environ["FEM_TILE_PACKED"] = env
    plans[name] = fa.Plan(om, fa.PREC_F32, part=part, ksteps=5 if scan else 7)
    plans[name].set_state(n0, c0)
bn = model.nMax
n_ref = bn + (n0 - bn).astype(np.float32).astype(np.float64)
c_ref = bn / (1 + bn) + (c0 - bn / (1 + bn)).astype(np.float32).astype(np.float64)
done = 0
for cp in checkpoints:
    n_ref, c_ref = st.step(n_ref, c_ref, dt, cp - done, model.as_array())
    out = {"steps": cp, "V": int(om.V), "max_dev_n": float(np.abs(n_ref - bn).max())}
    for name, plan in plans.items():
        plan.run(cp - done, dt, model)
        n, c = plan.get_state()
        out["rel_l2_n_" + name] = rel(n, n_ref)
        out["rel_l2_dev_" + name] = float(np.linalg.norm(n - n_ref) / np.linalg.norm(n_ref - bn))
    a, b = plans["packed"].get_state()[0], plans["fp32"].get_state()[0]
    out["rel_l2_packed_vs_fp32"] = rel(a, b)
    done = cp
    print(json.dumps(out), flush=True)
