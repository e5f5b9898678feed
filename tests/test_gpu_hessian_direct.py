"""The exact-Hessian kernel without LDS staging of its entries (hipnlp.hip, DIRECT instantiation: planar terrain, batch launches into
DEVICE memory — every entry goes straight to its place in the value run, four workgroups per CU) against the staged kernels: the same
bits, entry for entry, whatever the launch shape — periodicity as a cost (the 84 coupling entries the last knot writes behind its
block), shard handles (block offsets), non-finite iterates (NaN / Inf arrive in the run as from the staged kernel)."""
import os

import numpy as np
import pytest

from hippopt_amd import _abi
from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings
from hippopt_amd.synthetic import make_workload
from diag_lib import diag_library, diag_overrides

pytestmark = pytest.mark.gpu


def handle(HipNlp, st, model, B, direct, **kw):
    with diag_overrides(HIPNLP_HESS_DIRECT=1 if direct else 0) as lib:
        return HipNlp(st, model, batch=B, library=lib, **kw)


@pytest.mark.parametrize("case", ["periodic 100 x 8", "periodic 100 x 64 (the throughput leg: 6 400 workgroups, several rounds of four per CU)", "periodicity as a cost 9 x 70", "single step 30 x 20", "shard [3, 11) of 14 x 80", "non-finite iterates 12 x 50"])
def test_direct_store_hessian_kernel_is_the_staged_kernel_bit_for_bit(model, case):
    import torch
    from hippopt_amd.hipnlp import HipNlp
    kw = {}
    if case.startswith("periodic 100"):
        st, B = periodic_step_settings(100, model), (64 if "x 64" in case else 8)
    elif case.startswith("periodicity"):
        st, B = periodic_step_settings(9, model), 70
        st.periodicity_expression_type = _abi.EXPR_MINIMIZE
        st.periodicity_expression_weight = 3.5
    elif case.startswith("single"):
        st, B = single_step_settings(30, model), 20
    elif case.startswith("shard"):
        st, B = periodic_step_settings(14, model), 80
        kw = dict(knot_begin=3, knot_end=11)
    else:
        st, B = periodic_step_settings(12, model), 50
    N = st.horizon_length
    x, p = make_workload(st, model, batch=B, seed=9100 + N)
    rng = np.random.RandomState(9100 + B)
    x = x + 1e-2 * rng.standard_normal(x.shape)
    if case.startswith("non-finite"):
        x[7, 189 * 5 + 160] = np.nan          # a joint position of knot 5 (S_ + 3)
        x[31, 189 * 2 + 131] = np.inf        # a component of the base quaternion of knot 2 (QB_ + 1)
    direct, staged = handle(HipNlp, st, model, B, True, **kw), handle(HipNlp, st, model, B, False, **kw)
    assert direct.hess_nnz() == staged.hess_nnz() and (direct.desc.knot_end - direct.desc.knot_begin or N) * B > 512
    for e in (direct, staged):
        e.set_params(p)
    lam, sig = rng.standard_normal((B, direct.m)), rng.uniform(0.2, 2.0, B)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        xd, ld, sd = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev), torch.from_numpy(sig).to(dev)
        outs = [torch.full((B, direct.hess_nnz()), float("nan"), dtype=torch.float64, device=dev) for _ in range(2)]
    stream.synchronize()
    for rep in range(2):         # (the second launch finds the first one's values in place: nothing but this launch's stores may matter)
        for e, o in zip((direct, staged), outs):
            if rep == 1:
                o.fill_(123.0)
            e.eval_hess_device(xd.data_ptr(), sd.data_ptr(), ld.data_ptr(), o.data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()
        a, b = (o.cpu().numpy().view(np.int64) for o in outs)
        assert np.array_equal(a, b), (case, rep, int((a != b).sum()))
    if not case.startswith("non-finite"):     # ... and the host path (its own staged launch into the pinned block)
        assert np.array_equal(outs[0].cpu().numpy(), staged.eval_hess(x, sig, lam))
    else:
        assert np.isnan(outs[0].cpu().numpy()).any()
    for e in (direct, staged):
        e.close()
