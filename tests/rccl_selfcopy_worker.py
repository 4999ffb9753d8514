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

# Stream order of the exchange (fem_amd/dist.py:exchange_halo / wait_exchange): pack and consumers on a
# NON-current side stream, the pack delayed behind a long sleep kernel.  RCCL must not read the send buffer
# before the pack has run, and a consumer on the side stream must not read the ghost rows before the
# exchange has landed.
from fem_amd.dist import exchange_halo, wait_exchange  # noqa: E402


class _Shard:
    pass


class _Stepper:
    """pack = delayed copy on the given stream; ghost segment = the last m rows of the plan's state buffer"""

    def __init__(self, plan, V, m, typestr, dtype):
        self.full = torch.as_tensor(_DevView(plan.state_buffer(0), (V, 2), typestr), device="cuda")
        self.sendbuf = torch.zeros((m, 2), dtype=dtype, device="cuda")
        self.src = (torch.arange(2 * m, device="cuda", dtype=dtype).reshape(m, 2) + 3) / 512
        self.m, self.V = m, V

    def pack(self, stream, which=0):
        with torch.cuda.stream(stream):
            torch.cuda._sleep(200_000_000)           # ~0.1 s: the send buffer is still zero while this runs
            self.sendbuf.copy_(self.src)
        return self.sendbuf

    def ghost_view(self, which=0):
        return self.full[self.V - self.m:]


plan = fa.Plan(om, fa.PREC_F64)
plan.set_state(*fa.initial_condition(om.V, 5))
m = 53
st, sh = _Stepper(plan, om.V, m, "<f8", torch.float64), _Shard()
sh.send_counts, sh.recv_counts = [m], [m]
side = torch.cuda.Stream()
assert side != torch.cuda.current_stream()
for async_op in (True, False):
    st.sendbuf.zero_()
    st.ghost_view().zero_()
    torch.cuda.synchronize()
    work = exchange_halo(st, sh, side, async_op=async_op)
    wait_exchange(work if async_op else None, side)
    with torch.cuda.stream(side):
        got = st.ghost_view().clone()                # a consumer on the side stream
    torch.cuda.synchronize()
    assert torch.equal(got, st.src), "exchange on a side stream delivered stale rows (async_op=%s)" % async_op
print("selfcopy ok")
dist.barrier(device_ids=[0])
dist.destroy_process_group()
