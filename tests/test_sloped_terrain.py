"""Sloped step tops — SURVEY E16's  z_t = exp(-g^(2s)) pi(q_xy)  with the top surface of SmoothTerrain._top_expression_from_normal
(/root/reference/src/hippopt/robot_planning/utilities/smooth_terrain.py:238-264):  pi = height - (n_x q_x + n_y q_y) / n_z for a step
created with `top_normal_direction` (the ramp of turnkey_planners/humanoid_kinodynamic/main_walking_on_ramp.py:18-30, 403-409), the
height itself otherwise.  CPU side: the knot / pose programs the kernels compile (host emulation) against the forward-AD oracle — callback
quartet and exact Hessian — and the refusals of the reference's constructor at the C-ABI.  GPU side: tests/test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

from hippopt_amd import _abi, hipnlp
from hippopt_amd.kinodyn_settings import periodic_step_settings, ramp_settings, stairs_settings
from hippopt_amd.synthetic import make_workload, place_on_step_flanks
from hostemu_lib import HostEmu, PoseHostEmu
from oracle_lib import Oracle, PoseOracle

TOL = 1e-11


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(1.0, np.abs(np.asarray(b)))))


def terrains(horizon, model):
    ramp = ramp_settings(horizon, model)                                  # main_walking_on_ramp.py: one long step, normal (-0.2, 0, 1)
    mixed = stairs_settings(horizon, model)                               # a flat step and a sloped, rotated, offset, differently sharp one
    mixed.terrain_steps[1].update(orientation=-0.7, position=(0.8, 0.2, 0.03), edge_sharpness=3, side_sharpness=4,
                                  top_normal_direction=(0.3, -0.15, 2.0))
    return {"ramp": ramp, "mixed": mixed}


def test_ramp_settings_are_the_scripts():
    st = ramp_settings(50, None)
    (step,) = st.terrain_steps
    assert step["length"] == 0.9 and step["width"] == 0.8 and step["height"] == 0.1 and step["position"] == (0.675, 0.0, 0.0)
    assert step["top_normal_direction"] == (-0.2, 0.0, 1.0)
    c = st.to_c()
    assert c.n_terrain_steps == 1 and list(c.terrain_steps[0].top_normal) == [-0.2, 0.0, 1.0] and list(c.terrain_steps[1].top_normal) == [0.0, 0.0, 0.0]
    assert list(stairs_settings(5, None).to_c().terrain_steps[0].top_normal) == [0.0, 0.0, 0.0]       # None = the flat top


@pytest.mark.parametrize("name", ["ramp", "mixed"])
@pytest.mark.parametrize("horizon", [2, 4])
def test_sloped_tops_body_matches_oracle(model, name, horizon):
    st = terrains(horizon, model)[name]
    o, e = Oracle(st, model), HostEmu(st, model)
    assert (o.n, o.m, o.nnz) == (e.n, e.m, e.nnz) and o.row_blocks() == e.row_blocks()
    flat = stairs_settings(horizon, model)
    for flank in (True, False):
        x, p = make_workload(st, model, 1, 8100 + horizon)
        if flank:
            place_on_step_flanks(x, st, seed=horizon)
        f, grad, g, jac = o.eval(x[0], p[0])
        f2, grad2, g2, jac2, ct = e.eval(x[0], p[0])
        assert not np.isnan(g2).any() and not np.isnan(jac2).any()
        assert rel(f2, f) < 1e-9 and rel(grad2, grad) < 1e-9 and rel(g2, g) < TOL and rel(jac2, jac) < 1e-9
        assert np.allclose(ct, o.cost_terms(), rtol=1e-10, atol=1e-9)
        if name == "ramp":      # the slope is really there: the same step with a flat top gives other rows
            flat.terrain_steps = [dict(st.terrain_steps[0], top_normal_direction=None)]
            gf = Oracle(flat, model).eval(x[0], p[0])[2]
            assert np.max(np.abs(gf - g)) > 1e-3
    # closed form on the top of the ramp, far from its flanks: exp(-g^r) = 1 to machine precision, h = p_z - (o_z + height + slope q_x)
    if name == "ramp":
        step = st.terrain_steps[0]
        n = np.asarray(step["top_normal_direction"]) / np.linalg.norm(step["top_normal_direction"])
        x, p = make_workload(st, model, 1, 8200)
        from hippopt_amd.kinodyn_layout import variable_names
        at = {nm: off for nm, off, _ in variable_names(horizon)}["system[1].contact_points.left[0].p"]      # (the height rows start at knot 1)
        x[0, at:at + 3] = (step["position"][0] + 0.05, 0.02, 0.4)
        blocks = {b[0]: b for b in e.row_blocks()}
        first, rows, k0, nk = [b for nm, b in blocks.items() if nm.endswith("_height")][0][1:]
        want = 0.4 - (step["position"][2] + step["height"] - n[0] / n[2] * 0.05 - n[1] / n[2] * 0.02)
        assert k0 == 1 and abs(e.eval(x[0], p[0])[2][first] - want) < 1e-12


@pytest.mark.parametrize("name", ["ramp", "mixed"])
def test_sloped_tops_hessian_body_matches_oracle(model, name):
    from hess_util import hess_mismatch, triplets_to_dict
    st = terrains(3, model)[name]
    o, e = Oracle(st, model), HostEmu(st, model)
    x, p = make_workload(st, model, 1, 8300)
    place_on_step_flanks(x, st, seed=3)
    lam = np.random.RandomState(4).standard_normal(o.m)
    ref = triplets_to_dict(*o.hess(x[0], p[0], 0.7, lam))
    hr, hc = e.hess_sparsity()
    err, where = hess_mismatch(triplets_to_dict(hr, hc, e.hess(x[0], p[0], 0.7, lam)), ref, diag_scaled=True)
    assert err < 1e-9, where


def test_sloped_tops_in_the_pose_program(model):
    from hippopt_amd.pose_settings import make_pose_workload, pose_finder_settings
    st = pose_finder_settings(model)
    st.terrain = _abi.TERRAIN_SMOOTH_STEPS
    st.terrain_steps = [{"length": 0.9, "width": 0.8, "height": 0.1, "position": (0.3, 0.0, 0.0), "top_normal_direction": (-0.2, 0.0, 1.0)},
                        {"length": 0.3, "width": 0.5, "height": 0.1, "position": (-0.2, 0.1, 0.02), "orientation": 0.6, "edge_sharpness": 3, "side_sharpness": 4,
                         "top_normal_direction": (0.1, 0.2, 1.5)}]
    o, e = PoseOracle(st, model), PoseHostEmu(st, model)
    x, p = make_pose_workload(st, model, batch=1, seed=8400)
    f, grad, g, jac = o.eval(x[0], p[0])
    f2, grad2, g2, jac2, _ = e.eval(x[0], p[0])
    assert rel(f2, f) < 1e-9 and rel(grad2, grad) < 1e-9 and rel(g2, g) < TOL and rel(jac2, jac) < 1e-9
    flat = pose_finder_settings(model)
    flat.terrain, flat.terrain_steps = st.terrain, [dict(t, top_normal_direction=None) for t in st.terrain_steps]
    assert np.max(np.abs(PoseOracle(flat, model).eval(x[0], p[0])[2] - g)) > 1e-4
    from test_pose_body_hostemu import flank_points, hess_check
    flank_points(x[0], 4)
    lam = np.random.RandomState(5).standard_normal(o.m)
    hr, hc = e.hess_sparsity()
    vals = e.hess(x[0], p[0], 0.9, lam)
    assert not np.isnan(vals).any()
    hess_check(hr, hc, vals, o.hess(x[0], p[0], 0.9, lam), 1e-10)          # (the pose oracle returns the dense Hessian)


def test_top_normals_the_reference_refuses_are_refused_at_the_abi(model):
    """smooth_terrain.py:247-256: a top normal of (nearly) zero length, or parallel to the xy-plane, is a ValueError — HIPNLP_E_INVALID here,
    before any device is touched; the zero vector itself is the reference's None (flat top)"""
    lib = hipnlp.load_library()
    for normal, ok in (((0.0, 0.0, 0.0), True), ((-0.2, 0.0, 1.0), True), ((1e-9, 0.0, 0.0), False), ((1.0, 1.0, 0.0), False), ((0.3, 0.0, 1e-8), False)):
        st = ramp_settings(4, model)
        st.terrain_steps[0]["top_normal_direction"] = normal
        desc = _abi.DescC()
        desc.settings, desc.model, desc.batch = st.to_c(), model.to_c(), 1
        h = C.c_void_p()
        rc = lib.hipnlp_create(C.byref(desc), C.byref(h))
        message = lib.hipnlp_last_error(None)
        if h.value:
            lib.hipnlp_destroy(h)
        if ok:
            assert rc in (0, _abi.E_NODEVICE), (normal, rc, message)         # (accepted: the only thing that may be missing here is the device)
        else:
            assert rc == _abi.E_INVALID and b"top normal" in message, (normal, rc, message)
