#!/usr/bin/env python3
"""GPU box: ONE kinodynamic solve to convergence, with numbers.

The flow of main_single_step_flat_ground.py (:348-400): pose finder -> initial state (com (0, 0, h), feet side by side) and final
state (com 0.15 m ahead, right foot 0.3 m ahead), references (:325-345), then the planner.  The initial guess is the contact-phase
interpolation main_periodic_step.py:433-454 builds with `humanoid_state_interpolator` (the single-step script starts from the
dataclass defaults).  Settings: single step on flat ground, N = 30, dt = 0.1 (main_single_step_flat_ground.py:54-130).
Driver: IPOPT through cyipopt when importable, otherwise SciPy trust-constr (IPOPT is not in the image); both bind the engine's
callback quartet (and eval_h unless `hessian_approximation = limited-memory`).  detect_simple_bounds as in the reference's
casadi_opti_options.  Robot: the synthetic ergoCub-topology model (no URDF in the image).

Prints a JSON line: iterations, status, cost, constraint violation, callbacks per kind, new evaluations (kernel launches),
wall-clock split between the engine (inside the callbacks) and the driver.
  SOLVE_N=30 SOLVE_ITERS=3000 SOLVE_HESSIAN=exact|limited-memory python tools/diag/converged_solve.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hippopt_amd.robot_planning as hp_rp  # noqa: E402
from hippopt_amd import hipnlp_solver  # noqa: E402
from hippopt_amd.kinodyn_settings import single_step_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.robot_planning.transforms import SE3, SO3  # noqa: E402
from hippopt_amd.turnkey_planners import humanoid_pose_finder as pose_finder  # noqa: E402
from hippopt_amd.turnkey_planners.humanoid_kinodynamic import Planner, Settings  # noqa: E402
from hippopt_amd.turnkey_planners.humanoid_kinodynamic.variables import ExtendedHumanoidState, References  # noqa: E402

N = int(os.environ.get("SOLVE_N", "30"))
ITERS = int(os.environ.get("SOLVE_ITERS", "3000"))
HESSIAN = os.environ.get("SOLVE_HESSIAN", "exact")
COM_HEIGHT = float(os.environ.get("SOLVE_COM_HEIGHT", "0.62"))   # the synthetic robot is shorter than ergoCub (0.7 in the script)
DESIRED_JOINTS = np.deg2rad([7, 0.12, -0.01, 12.0, 7.0, -12.0, 40.769, 12.0, 7.0, -12.0, 40.769,
                             5.76, 1.61, -0.31, -31.64, -20.52, -1.52, 5.76, 1.61, -0.31, -31.64, -20.52, -1.52])


def pose(pf, settings, com, left_xyz, right_xyz):
    """compute_initial_state / compute_final_state of the script: the pose finder with foot and com references"""
    ref = pose_finder.References(contact_point_descriptors=settings.contact_points, number_of_joints=23)
    ref.state.com = np.asarray(com, float)
    ref.state.contact_points.left = hp_rp.FootContactState.from_parent_frame_transform(
        descriptor=settings.contact_points.left, transform=SE3.from_translation_and_rotation(np.asarray(left_xyz, float), SO3.Identity()))
    ref.state.contact_points.right = hp_rp.FootContactState.from_parent_frame_transform(
        descriptor=settings.contact_points.right, transform=SE3.from_translation_and_rotation(np.asarray(right_xyz, float), SO3.Identity()))
    ref.state.kinematics.base.quaternion_xyzw = np.array([0.0, 0.0, 0.0, 1.0])
    ref.frame_quaternion_xyzw = np.array([0.0, 0.0, 0.0, 1.0])
    ref.state.kinematics.joints.positions = DESIRED_JOINTS.copy()
    mass = pf.numeric_mass
    for pt in ref.state.contact_points.left + ref.state.contact_points.right:
        pt.f = np.array([0.0, 0.0, mass * 9.80665 / 8])
    pf.set_references(ref)
    guess = pf.get_initial_guess()
    guess.state.com = np.asarray(com, float)
    guess.state.kinematics.base.position = np.array([com[0], 0.0, COM_HEIGHT])
    guess.state.kinematics.joints.positions = DESIRED_JOINTS.copy()
    for pt, rf in zip(guess.state.contact_points.left + guess.state.contact_points.right,
                      ref.state.contact_points.left + ref.state.contact_points.right):
        pt.p, pt.f = np.asarray(rf.p, float).copy(), np.asarray(rf.f, float).copy()
    pf.set_initial_guess(guess)
    t0 = time.perf_counter()
    out = pf.solve()
    info = pf.optimization_solver._last_info
    return out.values.state, {"iterations": info.get("iterations"), "constr_violation": info.get("constr_violation"), "seconds": time.perf_counter() - t0}


def main():
    model = synthetic_ergocub()
    # ---- pose finder (get_pose_finder_settings, :135-170) ----------------------------------------------------------------------
    pst = pose_finder.Settings(solver_options={"max_iter": 300})
    pst.maximum_joint_positions = np.array(model.max_joint_positions, float)
    pst.minimum_joint_positions = np.array(model.min_joint_positions, float)
    pst.relaxed_complementarity_epsilon = 0.0001
    pst.static_friction = 0.3
    pst.base_quaternion_cost_multiplier = 50.0
    pst.desired_frame_quaternion_cost_multiplier = 100.0
    pst.joint_regularization_cost_multiplier = 0.1
    pst.force_regularization_cost_multiplier = 0.2
    pst.com_regularization_cost_multiplier = 10.0
    pst.average_force_regularization_cost_multiplier = 10.0
    pst.point_position_regularization_cost_multiplier = 100.0
    pf = pose_finder.Planner(pst, model, error_on_fail=False)
    initial, info_i = pose(pf, pst, (0.0, 0.0, COM_HEIGHT), (0.0, 0.1, 0.0), (0.0, -0.1, 0.0))
    print("pose finder, initial state:", info_i, file=sys.stderr, flush=True)
    final, info_f = pose(pf, pst, (0.15, 0.0, COM_HEIGHT), (0.0, 0.1, 0.0), (0.3, -0.1, 0.0))
    print("pose finder, final state:", info_f, file=sys.stderr, flush=True)

    # ---- planner settings (get_planner_settings, :17-130) -------------------------------------------------------------------------
    # the termination options of the reference script (main_single_step_flat_ground.py:105-130)
    opts = {"max_iter": ITERS, "tol": 1e-3, "constr_viol_tol": 1e-4, "acceptable_tol": 10.0, "acceptable_iter": int(os.environ.get("SOLVE_ACCEPTABLE_ITER", "2")),
            "acceptable_obj_change_tol": 1.0, "hessian_approximation": HESSIAN, "verbose": int(os.environ.get("SOLVE_VERBOSE", "0"))}
    st = Settings.from_numeric(single_step_settings(N, model), solver_options=opts)
    planner = Planner(st, model, error_on_fail=False)
    T = N * st.time_step
    ident = SO3.Identity()
    phases = hp_rp.FeetContactPhasesDescriptor()
    phases.left = [hp_rp.FootContactPhaseDescriptor(transform=SE3.from_translation_and_rotation(np.array([0.0, 0.1, 0.0]), ident),
                                                    force=np.array([0.0, 0.0, 100.0]))]
    phases.right = [
        hp_rp.FootContactPhaseDescriptor(transform=SE3.from_translation_and_rotation(np.array([0.0, -0.1, 0.0]), ident),
                                         mid_swing_transform=SE3.from_translation_and_rotation(np.array([0.15, -0.1, 0.05]), ident),
                                         force=np.array([0.0, 0.0, 100.0]), deactivation_time=T / 3.0),
        hp_rp.FootContactPhaseDescriptor(transform=SE3.from_translation_and_rotation(np.array([0.3, -0.1, 0.0]), ident),
                                         force=np.array([0.0, 0.0, 100.0]), activation_time=2.0 * T / 3.0)]
    guess_states = hp_rp.humanoid_state_interpolator(initial_state=initial, final_state=final, contact_phases=phases,
                                                     contact_descriptor=st.contact_points, number_of_points=N, dt=st.time_step)
    init_ext = ExtendedHumanoidState(contact_point_descriptors=st.contact_points, number_of_joints=23)
    init_ext.contact_points, init_ext.kinematics, init_ext.com = initial.contact_points, initial.kinematics, initial.com
    init_ext.centroidal_momentum = np.zeros(6)
    planner.set_initial_state(init_ext)
    planner.set_final_state(final)
    refs = References(number_of_joints=23, number_of_points_left=4, number_of_points_right=4)   # get_references, :325-345
    refs.contacts_centroid_cost_weights = np.array([100.0, 100.0, 10.0])
    refs.contacts_centroid = np.array([0.3, 0.0, 0.0])
    refs.joint_regularization = np.asarray(final.kinematics.joints.positions, float).reshape(-1)
    refs.com_linear_velocity = np.array([0.1, 0.0, 0.0])
    planner.set_references(refs)
    guess = planner.get_initial_guess()
    for k, state in enumerate(guess_states):
        sysk = guess.system[k]
        for dst, src in zip(sysk.contact_points.left + sysk.contact_points.right, state.contact_points.left + state.contact_points.right):
            dst.p, dst.f = np.asarray(src.p, float).reshape(-1), np.asarray(src.f, float).reshape(-1)
        sysk.kinematics.base.position = np.asarray(state.kinematics.base.position, float).reshape(-1)
        sysk.kinematics.base.quaternion_xyzw = np.asarray(state.kinematics.base.quaternion_xyzw, float).reshape(-1)
        sysk.kinematics.joints.positions = np.asarray(state.kinematics.joints.positions, float).reshape(-1)
        sysk.com = np.asarray(state.com, float).reshape(-1)
    planner.set_initial_guess(guess)

    # ---- solve, with the engine's share of the wall clock ---------------------------------------------------------------------------
    engine_s = [0.0]
    inner = hipnlp_solver._CallbackCache.eval

    def timed_eval(self, x, want, **kw):
        t = time.perf_counter()
        try:
            return inner(self, x, want, **kw)
        finally:
            engine_s[0] += time.perf_counter() - t
    hipnlp_solver._CallbackCache.eval = timed_eval
    sol = planner.optimization_solver
    # progress on stderr every 25 iterations (a silent GPU job is taken to be hung)
    inner_cb = sol._iterate_callback
    t_start = [time.perf_counter()]

    def progress(iteration, x, cost, inf_pr, multipliers, lam_x=None):
        if int(iteration) % 10 == 0:
            print("iteration %5d  cost %.6g  constraint violation %.3e  (%.1f s, engine %.2f s)" % (
                int(iteration), float(cost), float(inf_pr), time.perf_counter() - t_start[0], engine_s[0]), file=sys.stderr, flush=True)
        return inner_cb(iteration, x, cost, inf_pr, multipliers, lam_x)
    sol._iterate_callback = progress
    eng = sol.engine()
    hess_s, hess_n = [0.0], [0]
    eval_hess = eng.eval_hess

    def timed_hess(*a, **k):
        t = time.perf_counter()
        try:
            return eval_hess(*a, **k)
        finally:
            hess_s[0] += time.perf_counter() - t
            hess_n[0] += 1
    eng.eval_hess = timed_hess
    x0, p0 = sol._pack()
    eng.set_params(p0[None, :])
    _, _, g0, _ = eng.eval(x0[None, :])
    _, _, lbg, ubg = eng.bounds()
    viol0 = float(np.max(np.maximum(0, np.maximum(lbg - g0[0], g0[0] - ubg))))
    t0 = time.perf_counter()
    t_start[0] = t0
    out = planner.solve()
    wall = time.perf_counter() - t0
    info = sol._last_info
    res = {
        "problem": "kinodynamic single step on flat ground (main_single_step_flat_ground.py settings), N=%d, dt=%.2f, synthetic ergoCub-topology robot" % (N, st.time_step),
        "driver": "ipopt (cyipopt)" if "status_msg" in info else "scipy trust-constr (IPOPT absent)",
        "hessian": HESSIAN, "nlp": info.get("nlp"),
        "pose_finder": {"initial": info_i, "final": info_f},
        "iterations": info.get("iterations", info.get("iter_count")), "status": info.get("status"), "message": str(info.get("message"))[:120],
        "success": bool(info.get("success")),
        "cost": out.cost_value, "constraint_violation": info.get("constr_violation"), "constraint_violation_at_guess": viol0,
        "optimality": info.get("optimality"), "termination_options": {k: opts[k] for k in ("tol", "constr_viol_tol", "acceptable_tol", "acceptable_iter", "acceptable_obj_change_tol")},
        "callbacks": info.get("callbacks"), "hessian_evaluations": hess_n[0],
        "wall_s": wall, "engine_s": engine_s[0] + hess_s[0], "driver_s": wall - engine_s[0] - hess_s[0],
        "engine_us_per_callback_evaluation": 1e6 * engine_s[0] / max(1, (info.get("callbacks") or {}).get("evaluations", 1)),
        "engine_us_per_hessian": 1e6 * hess_s[0] / max(1, hess_n[0]),
    }
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
