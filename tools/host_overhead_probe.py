#!/usr/bin/env python3
"""CPU cost of one exchange period of the overlapped schedule (tiny shard: GPU time negligible)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fem_amd as fa
from fem_amd.dist import HipShardStepper

pts, tri = fa.make_icosphere(16, 0.1)
part = fa.Mesh(pts, tri).partition(12)
om = part.ordered
om.init_stats(fa.PREC_F64)
shard = fa.Shard(part, 0, 2, 8)
n0, c0 = fa.initial_condition(om.V, 1)
model = fa.default_model()
st = HipShardStepper(om, shard, "f32", model, n0[shard.loc2glob], c0[shard.loc2glob], part=part, ksteps=4)
nI, nB, nG = st.tile_classes()
stream = torch.cuda.current_stream()
dt = 1e-4
def period():
    st.launch(0, nI, 4, dt, model, stream); st.launch(nI, nB + nG, 4, dt, model, stream); st.flip()
    st.launch(nI, nB, 4, dt, model, stream); st.pack(stream, 1); st.ghost_view(1); st.launch(0, nI, 4, dt, model, stream); st.flip()
for _ in range(50): period()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 2000
for _ in range(N): period()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("classes", nI, nB, nG, "host us/period (4 launches + pack + 2 flips): %.1f   incl. drain: %.1f" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
