"""worker of test_sharded.py::test_rccl_writes_into_library_memory (launched by torch.distributed.run, world 1):
RCCL's all-to-all-v with the rank as its own peer, the receive buffer being a torch view of the
state buffer that libfem_hip.so allocated -- the data path of the halo exchange, with real bytes moving."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import fem_amd as fa  # noqa: E402
from fem_amd.dist import _DevView  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
pts, tri = fa.make_icosphere(8, 0.1)
mesh = fa.Mesh(pts, tri)
part = mesh.partition(4)
om = part.ordered
om.init_stats(fa.PREC_F64)
for prec, typestr, dtype in ((fa.PREC_F32, "<f4", torch.float32), (fa.PREC_F64, "<f8", torch.float64)):
    plan = fa.Plan(om, prec)
    n0, c0 = fa.initial_condition(om.V, 5)
    plan.set_state(n0, c0)
    m = 37
    full = torch.as_tensor(_DevView(plan.state_buffer(0), (om.V, 2), typestr), device="cuda")
    send = (torch.arange(2 * m, device="cuda", dtype=dtype).reshape(m, 2) + 1) / 1024
    work = dist.all_to_all_single(full[om.V - m:], send, [m], [m], async_op=True)
    assert hasattr(work, "is_completed")
    work.wait()
    torch.cuda.synchronize()
    assert work.is_completed()
    n, c = plan.get_state()
    base_n, base_c = 1.0, 0.5                      # the homogeneous state the buffers hold deviations from
    exp = send.cpu().numpy().astype(np.float64)
    assert np.abs(n[om.V - m:] - (base_n + exp[:, 0])).max() <= 1e-6
    assert np.abs(c[om.V - m:] - (base_c + exp[:, 1])).max() <= 1e-6
    tol = 1e-6 if prec == fa.PREC_F32 else 1e-15
    assert np.abs(n[:om.V - m] - n0[:om.V - m]).max() <= tol        # rows outside the receive window untouched
print("selfcopy ok")
dist.barrier(device_ids=[0])
dist.destroy_process_group()
