#!/usr/bin/env python3
"""Golden vectors of the static pose finder NLP (BASELINE config 2) obtained by EXECUTING THE REFERENCE'S OWN PYTHON here:
  hippopt.turnkey_planners.humanoid_pose_finder.planner.Planner.__init__   (planner.py:323-399 and everything it calls:
  OptimizationProblem.create, Problem.add_cost/add_constraint/add_expression, OptiSolver, robot_planning/expressions/*.py,
  utilities/planar_terrain.py / smooth_terrain.py / terrain_descriptor.py)
on the functional stand-ins of casadi / liecasadi / adam (tools/casadi_standin) and the synthetic robot.

Pins: variable / parameter creation order and names, constraint order / names / canonical bounds, cost scaling, every
formula as coded.  Does NOT pin CasADi's, adam's or liecasadi's own arithmetic (restated in the stand-ins).

Output: tests/golden/pose_<config>.npz.   Run:  python3 tools/gen_pose_fixtures.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from gen_planner_fixtures import EXPR, canon  # noqa: E402  (sets up sys.path for the stand-ins and the reference)

import numpy as np  # noqa: E402
import casadi as cs  # noqa: E402
import adam.casadi  # noqa: E402

import hippopt as hp  # noqa: E402
import hippopt.robot_planning as hp_rp  # noqa: E402
import hippopt.turnkey_planners.humanoid_pose_finder.planner as pose_finder  # noqa: E402

from hippopt_amd import _abi  # noqa: E402
from hippopt_amd.pose_settings import make_pose_workload, pose_finder_settings  # noqa: E402
from hippopt_amd.robot_model import JOINT_NAMES, synthetic_ergocub  # noqa: E402


def reference_settings(mine):
    s = pose_finder.Settings()
    s.robot_urdf = "synthetic://ergocub-topology"
    s.joints_name_list = list(JOINT_NAMES)
    s.root_link = "root_link"
    s.desired_frame_quaternion_cost_frame_name = "chest"
    s.contact_points = hp_rp.FeetContactPointDescriptors()
    s.contact_points.left = hp_rp.ContactPointDescriptor.rectangular_foot("l_sole", 0.232, 0.1, np.array([0.116, 0.05, 0.0]))
    s.contact_points.right = hp_rp.ContactPointDescriptor.rectangular_foot("r_sole", 0.232, 0.1, np.array([0.116, 0.05, 0.0]))
    for k in ("relaxed_complementarity_epsilon", "static_friction", "maximum_joint_positions", "minimum_joint_positions",
              "joint_regularization_cost_weights", "base_quaternion_cost_multiplier", "desired_frame_quaternion_cost_multiplier",
              "joint_regularization_cost_multiplier", "force_regularization_cost_multiplier", "com_regularization_cost_multiplier",
              "average_force_regularization_cost_multiplier", "point_position_regularization_cost_multiplier"):
        setattr(s, k, getattr(mine, k))
    s.com_position_expression_type = EXPR[mine.com_position_expression_type]
    s.left_point_position_expression_type = EXPR[mine.left_point_position_expression_type]
    s.right_point_position_expression_type = EXPR[mine.right_point_position_expression_type]
    # hand position expressions (planner.py:62-69): frame NAMES on the reference side
    s.left_hand_expression_type = EXPR[mine.left_hand_expression_type]
    s.right_hand_expression_type = EXPR[mine.right_hand_expression_type]
    s.left_hand_frame_name = getattr(mine, "left_hand_frame_name", None)
    s.right_hand_frame_name = getattr(mine, "right_hand_frame_name", None)
    s.lef_hand_position_in_frame = np.asarray(mine.lef_hand_position_in_frame, float)
    s.right_hand_position_in_frame = np.asarray(mine.right_hand_position_in_frame, float)
    s.left_hand_regularization_cost_multiplier = mine.left_hand_regularization_cost_multiplier
    s.right_hand_regularization_cost_multiplier = mine.right_hand_regularization_cost_multiplier
    if mine.terrain == _abi.TERRAIN_SMOOTH_STEPS:   # main_complex_poses.py:268-274
        terrain = None
        for st in mine.terrain_steps:
            step = hp_rp.SmoothTerrain.step(length=st["length"], width=st["width"], height=st["height"],
                                            position=np.array(st["position"], float), orientation=st.get("orientation", 0.0),
                                            edge_sharpness=st.get("edge_sharpness", 5), side_sharpness=st.get("side_sharpness", 10))
            terrain = step if terrain is None else terrain + step
        s.terrain = terrain
    s.casadi_function_options = {}
    s.casadi_opti_options = {}
    s.casadi_solver_options = {}
    assert s.is_valid()
    return s


def generate(tag, mine, model, seed, tweak=None):
    adam.casadi.STANDIN_MODEL = model
    planner = pose_finder.Planner(reference_settings(mine))
    solver = planner.optimization_solver
    opti = solver._solver
    x, p = make_pose_workload(mine, model, 1, seed)
    x, p = x[0], p[0]
    if tweak is not None:
        tweak(x)
    nx = sum(v.numel() for v in opti.variables)
    npar = sum(q.numel() for q in opti.parameters)
    assert nx == x.size and npar == p.size, (nx, x.size, npar, p.size)
    # flattened names / sizes / kinds in the reference's creation order
    values_dict, meta = planner.get_variables_structure().to_dicts()
    vnames, pnames = [], []
    for name, val in values_dict.items():
        size = int(np.asarray(val, dtype=float).size)
        (vnames if meta[name][hp.OptimizationObject.StorageTypeField] == "variable" else pnames).append("%s:%d" % (name, size))
    values, seeds, off = {}, {}, 0
    for v in opti.variables:
        k = v.numel()
        values[v._id] = x[off:off + k].reshape(v.shape, order="F")
        sd = np.zeros(v.shape + (nx,))
        for i in range(k):
            sd[i % v.shape[0], i // v.shape[0], off + i] = 1.0
        seeds[v._id] = sd
        off += k
    off = 0
    for q in opti.parameters:
        k = q.numel()
        values[q._id] = p[off:off + k].reshape(q.shape, order="F")
        off += k
    var_ids = {v._id for v in opti.variables}
    names, rows, gs, lbs, ubs = [], [], [], [], []
    for name, c in solver.get_constraint_expressions().items():
        g, lb, ub = canon(c, var_ids)
        gs.append(g)
        ones = cs.DM(np.ones(g.shape))
        lbs.append(lb * ones)
        ubs.append(ub * ones)
        names.append(name)
        rows.append(g.numel())
    cost_names = list(solver.get_cost_expressions().keys())
    cost_exprs = list(solver.get_cost_expressions().values())
    f_expr = solver.cost_function()
    G = cs.vertcat(*gs)
    (gv, fv), (gt, ft) = cs.evaluate([G, f_expr], values, seeds, nx)
    (lbv, ubv), _ = cs.evaluate([cs.vertcat(*lbs), cs.vertcat(*ubs)], values)
    cvals, _ = cs.evaluate(cost_exprs, values)
    J = gt[:, 0, :]
    grad = ft[0, 0, :]
    ir, jc = np.nonzero(J)
    order = np.lexsort((ir, jc))
    ir, jc = ir[order], jc[order]
    # Hessian of the Lagrangian sigma f + lambda^T g (what nlp_hess_l gives IPOPT's eval_h: the pose finder runs IPOPT with
    # the exact Hessian, humanoid_pose_finder/main.py:101): symbolic forward derivative of L along every variable component
    # (cs.jtimes on the reference's own graph), then its numeric tangents with respect to all variables.
    rng = np.random.RandomState(seed + 1)
    sigma = 0.75
    lam = rng.standard_normal(int(G.shape[0]))
    lag = sigma * f_expr + cs.mtimes(cs.DM(lam.reshape(1, -1)), G)
    rows_h = []
    for v in opti.variables:
        for i in range(v.numel()):
            e = np.zeros(v.shape)
            e[i % v.shape[0], i // v.shape[0]] = 1.0
            rows_h.append(cs.jtimes(lag, v, cs.DM(e)))
    _, ht = cs.evaluate(rows_h, values, seeds, nx)
    hess = np.stack([np.zeros(nx) if t is None else t[0, 0, :] for t in ht])
    assert np.max(np.abs(hess - hess.T)) < 1e-9 * max(1.0, np.max(np.abs(hess)))
    out = os.path.join(ROOT, "tests", "golden", "pose_%s.npz" % tag)
    np.savez_compressed(out, x=x, p=p, g=gv.reshape(-1), lbg=lbv.reshape(-1), ubg=ubv.reshape(-1), f=float(fv), grad=grad,
                        jac_row=ir.astype(np.int32), jac_col=jc.astype(np.int32), jac_val=J[ir, jc],
                        names=np.array(names), rows=np.array(rows, np.int32), cost_names=np.array(cost_names),
                        cost_values=np.array([float(np.asarray(c).reshape(-1)[0]) for c in cvals]),
                        vnames=np.array(vnames), pnames=np.array(pnames),
                        hess_sigma=sigma, hess_lambda=lam, hess=hess,
                        meta=np.array(json.dumps({"config": tag, "seed": seed,
                                                  "generator": "reference humanoid_pose_finder/planner.py on tools/casadi_standin (stand-in, not CasADi)"})))
    print(tag, "n", nx, "m", int(G.shape[0]), "nnz(numeric)", len(ir), "f", float(fv), "->", os.path.relpath(out, ROOT))


def step_settings(model):
    """main_complex_poses.py:268-274 shape: one high smooth step; com and left-foot positions as constraints."""
    st = pose_finder_settings(model)
    st.terrain = _abi.TERRAIN_SMOOTH_STEPS
    st.terrain_steps = [{"length": 0.6, "width": 0.8, "height": 0.2, "position": (0.45, 0.0, 0.0)}]
    st.com_position_expression_type = _abi.EXPR_SUBJECT_TO
    st.left_point_position_expression_type = _abi.EXPR_SUBJECT_TO
    st.right_point_position_expression_type = _abi.EXPR_SKIP
    return st


def hands_settings(model):
    """Both hand position expressions: the left hand as three equality rows, the right hand as a cost (planner.py:596-660)."""
    st = pose_finder_settings(model)
    st.left_hand_frame_name, st.right_hand_frame_name = "l_hand_palm", "r_hand_palm"
    st.left_hand_frame, st.right_hand_frame = model.resolve_frame("l_hand_palm"), model.resolve_frame("r_hand_palm")
    st.lef_hand_position_in_frame = np.array([0.01, 0.02, 0.03])
    st.right_hand_position_in_frame = np.array([0.0, -0.02, 0.05])
    st.left_hand_expression_type, st.right_hand_expression_type = _abi.EXPR_SUBJECT_TO, _abi.EXPR_MINIMIZE
    st.left_hand_regularization_cost_multiplier, st.right_hand_regularization_cost_multiplier = 0.7, 3.0
    st.right_point_position_expression_type = _abi.EXPR_SUBJECT_TO   # (rows BEHIND the hand rows: planner.py:385-399)
    return st


def on_the_flank(x):
    rng = np.random.RandomState(11)
    for c in range(8):
        a, b = rng.uniform(0.95, 1.01), rng.uniform(0.0, 0.9)
        if c % 3 == 0:
            a, b = b, rng.uniform(0.95, 1.01)
        x[6 * c + 0] = 0.45 + rng.choice([-1.0, 1.0]) * 0.3 * a
        x[6 * c + 1] = rng.choice([-1.0, 1.0]) * 0.4 * b
        x[6 * c + 2] = 0.1 + 0.05 * rng.standard_normal()


def main():
    model = synthetic_ergocub()
    generate("default", pose_finder_settings(model), model, 5001)
    generate("step_constrained", step_settings(model), model, 5002, tweak=on_the_flank)
    generate("hands", hands_settings(model), model, 5003)


if __name__ == "__main__":
    main()
