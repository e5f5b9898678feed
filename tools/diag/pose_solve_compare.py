#!/usr/bin/env python3
"""Diagnostic: the pose finder mirror solved with the exact Hessian of the Lagrangian (hipnlp_pose_eval_hess) and with the
quasi-Newton stand-in, same start, same driver (SciPy trust-constr; IPOPT is not in the image)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hippopt_amd.pose_settings import make_pose_workload
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.turnkey_planners.humanoid_pose_finder import Planner, References, Settings

model = synthetic_ergocub()
for mode in ("exact", "limited-memory"):
    st = Settings(solver_options={"max_iter": 150, "hessian_approximation": mode})
    st.maximum_joint_positions = np.array(model.max_joint_positions, float)
    st.minimum_joint_positions = np.array(model.min_joint_positions, float)
    pl = Planner(st, model, error_on_fail=False)
    mass = model.get_total_mass()
    x, _ = make_pose_workload(st, model, 1, 42)
    refs = References(contact_point_descriptors=st.contact_points, number_of_joints=23)
    refs.state.com = x[0][78:81].copy()
    for c, pt in enumerate(refs.state.contact_points.left + refs.state.contact_points.right):
        pt.p = x[0][6 * c:6 * c + 3].copy(); pt.p[2] = 0.0
        pt.f = np.array([0.0, 0.0, mass * 9.80665 / 8])
    refs.state.kinematics.joints.positions = x[0][55:78].copy()
    pl.set_references(refs)
    guess = pl.get_initial_guess()
    for c, pt in enumerate(guess.state.contact_points.left + guess.state.contact_points.right):
        pt.p = x[0][6 * c:6 * c + 3].copy(); pt.f = x[0][6 * c + 3:6 * c + 6] * mass
    guess.state.kinematics.base.position = x[0][48:51].copy()
    guess.state.kinematics.base.quaternion_xyzw = x[0][51:55].copy()
    guess.state.kinematics.joints.positions = x[0][55:78].copy()
    guess.state.com = x[0][78:81].copy()
    pl.set_initial_guess(guess)
    t0 = time.perf_counter()
    out = pl.solve()
    info = pl.optimization_solver._last_info
    print("%-15s iterations %4d  cost %.6f  constraint violation %.3e  status %s  (%.2f s)" % (
        mode, info.get("iterations", -1), out.cost_value, info.get("constr_violation", float("nan")), info.get("status"), time.perf_counter() - t0))
