"""The cut of a multi-device handle (hipnlp_multi_plan, no device needed): knot ranges as hippopt_amd.sharded.knot_range, and the x a
shard reads — its own records, the one-knot halo in front of them (the trapezoid defect of interval k - 1 -> k is a row of knot k:
/root/reference/src/hippopt/base/multiple_shooting_solver.py:713-742), the six horizon-global variables, the record of the other horizon
end for the owners of knot 0 and knot N - 1 (periodicity rows, turnkey_planners/humanoid_kinodynamic/planner.py:897-930).

Checked on the host emulation of the knot program (tests/hostemu: the task bodies the kernels compile): every shard evaluates its knots
on an x that is NaN everywhere outside its ranges; the shards' outputs tile the whole problem — every entry of grad f, g and jac g
written exactly once — and are, bit for bit, the evaluation of the whole horizon."""
import ctypes as C

import numpy as np
import pytest

from hippopt_amd import _abi
from hippopt_amd.hipnlp import multi_plan
from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings, stairs_settings
from hippopt_amd.sharded import knot_range
from hippopt_amd.synthetic import make_workload
from hostemu_lib import HostEmu, _dp


@pytest.mark.parametrize("horizon,shards", [(100, 4), (100, 8), (30, 7), (7, 7), (5, 2), (2, 2), (2, 1), (101, 3)])
def test_the_cut_is_contiguous_balanced_and_sharded_pys(horizon, shards):
    at = 0
    for i in range(shards):
        (kb, ke), ranges = multi_plan(horizon, shards, i)
        assert (kb, ke) == knot_range(horizon, shards, i) and kb == at and ke > kb
        at = ke
        first = ranges[0]
        assert first == (189 * max(kb - 1, 0), 189 * (ke - max(kb - 1, 0)))          # own records + the halo
        assert (189 * horizon, 6) in ranges                                          # the horizon-global variables
        other = [r for r in ranges[2:]]
        want = []
        if kb == 0 and ke < horizon:
            want.append((189 * (horizon - 1), 189))
        if ke == horizon and kb > 1:
            want.append((0, 189))
        assert other == want
    assert at == horizon
    from hippopt_amd.hipnlp import HipNlpError
    for bad in ((horizon, horizon + 1, 0), (horizon, shards, shards), (horizon, 0, 0), (1, 1, 0)):
        with pytest.raises(HipNlpError):
            multi_plan(*bad)


@pytest.mark.parametrize("maker,horizon,shards", [(periodic_step_settings, 12, 4), (periodic_step_settings, 9, 9), (single_step_settings, 10, 3),
                                                  (stairs_settings, 8, 2), (periodic_step_settings, 6, 1)])
@pytest.mark.parametrize("periodicity_as_cost", [False, True])
def test_shards_that_read_only_their_ranges_tile_the_whole_evaluation(model, maker, horizon, shards, periodicity_as_cost):
    st = maker(horizon, model)
    if periodicity_as_cost:
        st.periodicity_expression_type = _abi.EXPR_MINIMIZE
        st.final_state_expression_type = _abi.EXPR_MINIMIZE
    x, p = make_workload(st, model, batch=1, seed=7600 + horizon)
    x, p = x[0], p[0]
    he = HostEmu(st, model)
    f, grad, g, jac, ct = he.eval(x, p)
    assert np.all(np.isfinite(g)) and np.all(np.isfinite(jac))
    poison = np.nan
    G, Gg, J = np.full(he.n, poison), np.full(he.m, poison), np.full(he.nnz, poison)
    hits = [np.zeros(he.n, int), np.zeros(he.m, int), np.zeros(he.nnz, int)]
    cost = np.full((horizon, _abi.NCOST_TERMS), poison)
    fn = he.lib.hostemu_eval_range
    for i in range(shards):
        (kb, ke), ranges = multi_plan(horizon, shards, i)
        xs = np.full_like(x, np.nan)                   # what the shard is given: its ranges and nothing else
        for off, cnt in ranges:
            xs[off:off + cnt] = x[off:off + cnt]
        mine = [np.full(he.n, poison), np.full(he.m, poison), np.full(he.nnz, poison)]
        fn(C.c_void_p(he.h), _dp(xs), _dp(np.ascontiguousarray(p)), kb, ke, _dp(mine[0]), _dp(mine[1]), _dp(mine[2]), _dp(cost))
        for dst, src, h in zip((G, Gg, J), mine, hits):
            wrote = ~np.isnan(src)
            dst[wrote] = src[wrote]
            h += wrote
    for name, got, ref, h in zip(("grad", "g", "jac"), (G, Gg, J), (grad, g, jac), hits):
        assert np.all(h == 1), (name, int((h == 0).sum()), int((h > 1).sum()))       # every entry by exactly one shard
        assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), name
    assert np.all(np.isfinite(cost))
    assert np.allclose(cost.sum(axis=0), ct, rtol=1e-14, atol=1e-14)


def test_multi_create_refuses_bad_arguments_before_it_needs_a_device(model):
    """hipnlp_multi_create validates the cut first (no device needed to be told); with a valid request and no device it fails as loudly as
    hipnlp_create does — there is no CPU path behind either"""
    import torch
    from hippopt_amd import hipnlp
    lib = hipnlp.load_library()
    desc = _abi.DescC()
    desc.settings, desc.model, desc.batch = periodic_step_settings(4, model).to_c(), model.to_c(), 1
    h = C.c_void_p()

    def create(devices, d=desc):
        arr = (C.c_int32 * max(1, len(devices)))(*devices)
        rc = lib.hipnlp_multi_create(C.byref(d), arr, len(devices), C.byref(h))
        return rc, lib.hipnlp_last_error(None).decode()
    rc, msg = create([])
    assert rc == _abi.E_INVALID and "1 .. 64" in msg and not h.value
    rc, msg = create([0] * 5)                       # five shards, four knots
    assert rc == _abi.E_INVALID and "more devices than knots" in msg
    part = _abi.DescC.from_buffer_copy(bytes(desc))
    part.knot_begin, part.knot_end = 1, 3           # the library cuts the horizon itself
    rc, msg = create([0, 0], part)
    assert rc == _abi.E_INVALID and "whole horizon" in msg
    old = _abi.DescC.from_buffer_copy(bytes(desc))
    old.abi_version = _abi.ABI_VERSION - 1
    rc, msg = create([0, 0], old)
    assert rc == _abi.E_INVALID and "abi_version" in msg
    if not torch.cuda.is_available():
        rc, msg = create([0, 0])
        assert rc == _abi.E_NODEVICE and "no CPU fallback" in msg and not h.value
    n = C.c_int32(4)
    assert lib.hipnlp_multi_info(None, C.byref(n), None, None, None, None) == _abi.E_INVALID
