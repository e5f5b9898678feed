"""The pose program (hippopt_amd/csrc/pose_body.h) and the pose layout tables through the TEST-ONLY host emulation, against
the AD oracle and the golden vectors of the reference's pose finder.  GPU parity proper: test_gpu_pose.py."""
import json
import os

import numpy as np
import pytest

from hippopt_amd import _abi
from hippopt_amd.pose_settings import make_pose_workload, pose_finder_settings
from hostemu_lib import PoseHostEmu
from oracle_lib import PoseOracle
from test_golden_pose import GOLD, check_against_fixture, pose_settings_for

TOL = 1e-11


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def variants(model):
    a = pose_finder_settings(model)
    b = pose_finder_settings(model)
    b.com_position_expression_type = _abi.EXPR_SUBJECT_TO
    b.left_point_position_expression_type = _abi.EXPR_SKIP
    b.right_point_position_expression_type = _abi.EXPR_SUBJECT_TO
    c = pose_finder_settings(model)
    c.terrain = _abi.TERRAIN_SMOOTH_STEPS
    c.terrain_steps = [{"length": 0.6, "width": 0.8, "height": 0.2, "position": (0.45, 0.0, 0.0)},
                       {"length": 0.3, "width": 0.5, "height": 0.1, "position": (-0.2, 0.1, 0.02), "orientation": 0.6, "edge_sharpness": 3, "side_sharpness": 4}]
    c.com_position_expression_type = _abi.EXPR_SKIP
    # hand position expressions (planner.py:596-660): frames on the last arm links of the synthetic tree, one hand per mode
    from hippopt_amd.pose_settings import hand_frame
    d = pose_finder_settings(model)
    d.left_hand_frame = hand_frame(model, 7, (0.1, -0.2, 0.3), (0.02, 0.01, -0.05))
    d.right_hand_frame = hand_frame(model, 11, (0.0, 0.1, 0.0), (0.0, 0.0, -0.04))
    d.lef_hand_position_in_frame = np.array([0.01, 0.02, 0.03])
    d.left_hand_expression_type, d.right_hand_expression_type = _abi.EXPR_SUBJECT_TO, _abi.EXPR_MINIMIZE
    d.left_hand_regularization_cost_multiplier, d.right_hand_regularization_cost_multiplier = 0.7, 3.0
    e = pose_finder_settings(model)
    e.terrain, e.terrain_steps = c.terrain, c.terrain_steps
    e.left_hand_frame, e.right_hand_frame = d.left_hand_frame, d.right_hand_frame
    e.right_hand_position_in_frame = np.array([0.0, -0.02, 0.05])
    e.left_hand_expression_type, e.right_hand_expression_type = _abi.EXPR_MINIMIZE, _abi.EXPR_SUBJECT_TO
    e.left_hand_regularization_cost_multiplier = 12.0
    f = pose_finder_settings(model)      # both hands minimised (their gradient shares meet in p_b, q_b and the torso joints)
    f.left_hand_frame, f.right_hand_frame = d.left_hand_frame, d.right_hand_frame
    f.left_hand_expression_type = f.right_hand_expression_type = _abi.EXPR_MINIMIZE
    f.right_hand_regularization_cost_multiplier = 2.5
    return {"default": a, "constrained": b, "steps": c, "hands": d, "hands_steps": e, "hands_costs": f}


VARIANTS = ["default", "constrained", "steps", "hands", "hands_steps", "hands_costs"]


@pytest.mark.parametrize("name", VARIANTS)
def test_pose_body_matches_oracle(model, name):
    st = variants(model)[name]
    o, e = PoseOracle(st, model), PoseHostEmu(st, model)
    assert (o.n, o.m, o.nnz) == (e.n, e.m, e.nnz)
    assert o.row_blocks() == e.row_blocks()
    ir, jc = o.sparsity()
    ir2, jc2 = e.sparsity()
    assert np.array_equal(ir, ir2) and np.array_equal(jc, jc2)
    x, p = make_pose_workload(st, model, 3, 700)
    if name.endswith("steps"):   # contact points on the flanks of the first bump
        rng = np.random.RandomState(1)
        for c in range(8):
            x[0][6 * c] = 0.45 + rng.choice([-1.0, 1.0]) * 0.3 * rng.uniform(0.95, 1.01)
            x[0][6 * c + 1] = 0.3 * rng.uniform(-1, 1)
            x[0][6 * c + 2] = 0.1 + 0.05 * rng.standard_normal()
    for b in range(3):
        f, grad, g, jac = o.eval(x[b], p[b])
        f2, grad2, g2, jac2, ct = e.eval(x[b], p[b])
        assert not np.isnan(g2).any() and not np.isnan(jac2).any()
        assert rel(np.array(f2), np.array(f)) < TOL and rel(grad2, grad) < TOL and rel(g2, g) < TOL and rel(jac2, jac) < 1e-10
        assert np.allclose(ct, o.cost_terms(), rtol=1e-12, atol=1e-12)
        lb, ub = o.bounds(p[b])
        lb2, ub2 = e.bounds(p[b])
        assert np.array_equal(lb, lb2) and np.array_equal(ub, ub2)
    # the emulation runs the pose program on the DEVICE emitter's cut Jacobian staging (pose_body.h pjs): every emission was one the device can serve
    assert e.map_violations() == 0


@pytest.mark.parametrize("name", ["pose_default", "pose_step_constrained", "pose_hands"])
def test_pose_body_matches_reference_fixture(model, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    st = pose_settings_for(json.loads(str(z["meta"])), model)
    e = PoseHostEmu(st, model)
    f, grad, g, jac, _ = e.eval(z["x"], z["p"])
    lb, ub = e.bounds(z["p"])
    ir, jc = e.sparsity()
    check_against_fixture(z, ir, jc, f, grad, g, jac, lb, ub, tol=1e-10)


# ---- exact Hessian of the Lagrangian (pose_hess_body.h) -------------------------------------------------------------------
def hess_check(ir, jc, vals, Href, tol):
    """vals on the lower-triangle pattern (ir >= jc, column major) against a dense symmetric reference"""
    assert np.all(ir >= jc)
    assert np.array_equal(np.lexsort((ir, jc)), np.arange(ir.size))           # CCS order, no duplicates
    assert len(set(zip(ir.tolist(), jc.tolist()))) == ir.size
    H = np.zeros_like(Href)
    H[ir, jc] = vals
    L = np.tril(Href)
    scale = np.maximum(1.0, np.abs(L))
    assert np.max(np.abs(H - L) / scale) < tol
    inside = np.zeros_like(Href, bool)
    inside[ir, jc] = True
    assert np.max(np.abs(L[~inside]), initial=0.0) == 0.0                        # nothing outside the pattern


def flank_points(x, seed):
    rng = np.random.RandomState(seed)
    for c in range(8):
        x[6 * c] = 0.45 + rng.choice([-1.0, 1.0]) * 0.3 * rng.uniform(0.95, 1.01)
        x[6 * c + 1] = 0.3 * rng.uniform(-1, 1)
        x[6 * c + 2] = 0.1 + 0.05 * rng.standard_normal()


@pytest.mark.parametrize("name", VARIANTS)
def test_pose_hessian_body_matches_oracle(model, name):
    st = variants(model)[name]
    o, e = PoseOracle(st, model), PoseHostEmu(st, model)
    ir, jc = e.hess_sparsity()
    x, p = make_pose_workload(st, model, 3, 900)
    if name.endswith("steps"):
        flank_points(x[0], 4)
    rng = np.random.RandomState(5)
    pattern = np.zeros((o.n, o.n), bool)
    for b in range(3):
        lam, sigma = rng.standard_normal(o.m), float(rng.uniform(0.2, 2.0))
        Href = o.hess(x[b], p[b], sigma, lam)
        vals = e.hess(x[b], p[b], sigma, lam)
        assert not np.isnan(vals).any()
        hess_check(ir, jc, vals, Href, 1e-10 if name.endswith("steps") else TOL)
        pattern |= np.tril(Href) != 0.0
    # the pattern is tight: every entry of it is non-zero at some sample
    mine = np.zeros_like(pattern)
    mine[ir, jc] = True
    assert np.array_equal(mine, pattern)
    assert e.map_violations() == 0   # (the multipliers went in as the device kernel stages them: row i to the slot of row i)


@pytest.mark.parametrize("name", ["pose_default", "pose_step_constrained", "pose_hands"])
def test_pose_hessian_body_matches_reference_fixture(model, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    st = pose_settings_for(json.loads(str(z["meta"])), model)
    e = PoseHostEmu(st, model)
    ir, jc = e.hess_sparsity()
    vals = e.hess(z["x"], z["p"], float(z["hess_sigma"]), z["hess_lambda"])
    hess_check(ir, jc, vals, z["hess"], 1e-10)


@pytest.mark.parametrize("name", VARIANTS)
def test_pose_program_phases_are_wave_order_independent(model, name):
    """Groups of one phase on different waves run concurrently on the GPU: the pose program and its Hessian program, every phase wave by
    wave in three orders of the four waves (in-wave table order kept), must give the table-order values bit for bit (cross-wave
    dependencies inside a phase, two waves adding into one scratch entry)."""
    import hostemu_lib
    st = variants(model)[name]
    e = PoseHostEmu(st, model)
    x, p = make_pose_workload(st, model, 1, 1500)
    if name.endswith("steps"):
        flank_points(x[0], 4)
    lam = np.random.RandomState(7).standard_normal(e.m)
    ref, href = e.eval(x[0], p[0]), e.hess(x[0], p[0], 1.1, lam)
    try:
        for order in (0, 1, 2):
            hostemu_lib.set_wave_order(order)
            got, hgot = e.eval(x[0], p[0]), e.hess(x[0], p[0], 1.1, lam)
            assert got[0] == ref[0]
            for a, b, what in zip(got[1:], ref[1:], ("grad", "g", "jac", "cost terms")):
                assert np.array_equal(a, b), (order, what, int(np.argmax(a != b)))
            assert np.array_equal(hgot, href), (order, int(np.argmax(hgot != href)))
    finally:
        hostemu_lib.set_wave_order(-1)
