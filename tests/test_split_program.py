"""Workgroup specialisation of the knot program (hippopt_amd/csrc/knot_body.h, split_task_is_model_free; the SPLIT kernels of hipnlp.hip):
the task groups that need the robot model and the model-free ones run on two workgroups of one launch.  That is only right if no group
of one half reads scratch the other half writes and every output has one owner.  Checked without a GPU on the host emulation: every knot
by two passes over two NaN-poisoned scratches, one per half, every output taken from the half that owns it — bit for bit the one-scratch
run, nothing left unwritten.  Scheduling only: the rows and costs are those of the reference's list
(/root/reference/src/hippopt/turnkey_planners/humanoid_kinodynamic/planner.py:124-176)."""
import ctypes as C

import numpy as np
import pytest

from hippopt_amd import _abi
from hippopt_amd.kinodyn_settings import periodic_step_settings, ramp_settings, single_step_settings, stairs_settings
from hippopt_amd.synthetic import make_workload, place_on_step_flanks
from hostemu_lib import HostEmu, _dp


def split_eval(he, x, p):
    f = C.c_double()
    grad, g, jac, ct = np.zeros(he.n), np.full(he.m, np.nan), np.full(he.nnz, np.nan), np.zeros(_abi.NCOST_TERMS)
    he.lib.hostemu_eval_split.restype = C.c_int
    unwritten = he.lib.hostemu_eval_split(C.c_void_p(he.h), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(p)), C.byref(f), _dp(grad), _dp(g), _dp(jac), _dp(ct))
    return unwritten, f.value, grad, g, jac, ct


@pytest.mark.parametrize("maker,horizon", [(periodic_step_settings, 5), (single_step_settings, 4), (stairs_settings, 4), (ramp_settings, 3), (periodic_step_settings, 2)])
@pytest.mark.parametrize("lifted,vary_first", [(False, False), (True, True)])
def test_the_two_halves_share_nothing_and_own_every_output(model, maker, horizon, lifted, vary_first):
    st = maker(horizon, model)
    x, p = make_workload(st, model, 1, 8700 + horizon)
    if st.terrain != _abi.TERRAIN_PLANAR:
        place_on_step_flanks(x, st, seed=horizon)
    he = HostEmu(st, model, detect_simple_bounds=lifted, jac_varying_first=vary_first)
    f, grad, g, jac, ct = he.eval(x[0], p[0])
    unwritten, f2, grad2, g2, jac2, ct2 = split_eval(he, x[0], p[0])
    assert unwritten == 0
    for name, a, b in (("grad", grad, grad2), ("g", g, g2), ("jac", jac, jac2), ("cost terms", ct, ct2)):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), name
    assert f == f2


def test_horizon_ends_as_costs_are_why_such_settings_keep_one_workgroup(model):
    """final state / periodicity in `minimize` mode: t_ends_finish (model-free half) adds into EVERY gradient entry, the kinematic half's
    too — the split run then differs from the whole one in exactly those entries, which is why launch() never splits such settings"""
    st = periodic_step_settings(4, model)
    st.final_state_expression_type = st.periodicity_expression_type = _abi.EXPR_MINIMIZE
    x, p = make_workload(st, model, 1, 8800)
    he = HostEmu(st, model)
    f, grad, g, jac, ct = he.eval(x[0], p[0])
    _, f2, grad2, g2, jac2, ct2 = split_eval(he, x[0], p[0])
    assert np.array_equal(g, g2) and np.array_equal(jac, jac2) and np.array_equal(ct, ct2)
    differ = np.nonzero(grad != grad2)[0] % 189
    assert differ.size > 0 and np.all((differ >= 130) & (differ < 180))     # the kinematic half's entries of the first / last knot
