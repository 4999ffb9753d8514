"""fem_plan_format_noalloc (the entry point that allocates nothing) against fem_plan_format, through the C-ABI:
two calls in a row on one plan, the second on fewer vertices, bit for bit."""
import numpy as np
import pytest

import fem_amd as fa
from helpers import smooth_ic, sphere_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", [fa.PREC_F32, fa.PREC_F64])
def test_format_noalloc_matches_format(prec):
    import torch
    case = sphere_case(40, 16)                       # V = 16 002: 63 partial blocks
    case.mesh.init_stats(fa.PREC_F64)
    plan = fa.Plan(case.mesh, prec, part=case.part, ksteps=4)
    n0, c0 = smooth_ic(case.pts)
    plan.set_state(n0, c0)
    plan.run(9, case.dt, case.model)
    V = plan.V
    out = torch.full((V, 4), -7.0, dtype=torch.float64, device="cuda")       # the caller's buffers
    mm = torch.zeros(2, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for nverts in (V, 5000, 300):                   # second and third call smaller: the array keeps the first call's partials
        rgba_ref, _, lo_ref, hi_ref = plan.format(nverts)
        out.fill_(-7.0)
        plan.format_noalloc(out.data_ptr(), mm.data_ptr(), nverts, stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got[:nverts], rgba_ref)
        assert (got[nverts:] == -7.0).all()         # nothing written beyond nverts
        lo, hi = mm.cpu().numpy()
        assert lo == lo_ref and hi == hi_ref
    # without the extrema output
    out.fill_(-7.0)
    plan.format_noalloc(out.data_ptr(), 0, V, stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), plan.format(V)[0])


def test_format_noalloc_rejects_bad_arguments():
    import torch
    case = sphere_case(8, 4)
    case.mesh.init_stats(fa.PREC_F64)
    plan = fa.Plan(case.mesh, fa.PREC_F32)
    out = torch.zeros((plan.V, 4), dtype=torch.float64, device="cuda")
    for args in ((0, 0, plan.V + 1), (out.data_ptr() + 8, 0, plan.V), (out.data_ptr(), 4, plan.V)):
        with pytest.raises(fa.FemError):
            plan.format_noalloc(args[0] if args[0] else out.data_ptr(), args[1], args[2])
    # the size limit of the fixed array is checked before anything is launched
    assert fa.lib.fem_plan_format_noalloc(plan.h, (1 << 26) + 1, out.data_ptr(), 0, None) < 0
