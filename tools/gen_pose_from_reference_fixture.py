#!/usr/bin/env python3
"""hippopt_amd.from_reference.pose_from_reference on the REFERENCE'S OWN pose-finder objects, in this container.

Imports the reference (hippopt, /root/reference/src) with the inert stubs under tools/refstub and builds, with the reference's own
classes (turnkey_planners/humanoid_pose_finder/planner.py), a `Settings` filled the way humanoid_pose_finder/main.py:57-101 fills
it plus both hand position expressions (frame NAMES, :62-69) and a `Variables(settings, kin_dyn_object)` (:229-320) filled
through the reference's own `OptimizationObject.from_dict` with a seeded pose in PHYSICAL units (contact forces of the state and of
the references multiplied by the mass), hands both to pose_from_reference and stores (hipnlp_pose_desc bytes, x, p) in
tests/golden/pose_from_reference.npz.  The tests replay the fixture: the bytes must equal what the build's own settings / model
classes produce, x and p the seeded pose, and on the GPU the engine created from the stored bytes must agree with the oracle.
Nothing of the reference travels: only the arrays do.

Run:  python3 tools/gen_pose_from_reference_fixture.py
"""
import ctypes
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "refstub"))
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import hippopt as hp  # noqa: E402
import hippopt.robot_planning as hp_rp  # noqa: E402
import hippopt.turnkey_planners.humanoid_pose_finder.planner as pf  # noqa: E402

from hippopt_amd import _abi  # noqa: E402
from hippopt_amd.from_reference import pose_from_reference  # noqa: E402
from hippopt_amd.pose_settings import make_pose_workload, pose_finder_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.urdf_model import to_urdf  # noqa: E402

SEED = 6161


def numeric_settings(model):
    st = pose_finder_settings(model)
    st.left_hand_frame, st.right_hand_frame = model.resolve_frame("l_hand_palm"), model.resolve_frame("r_hand_palm")
    st.lef_hand_position_in_frame = np.array([0.01, 0.02, 0.03])
    st.right_hand_position_in_frame = np.array([0.0, -0.02, 0.05])
    st.left_hand_expression_type, st.right_hand_expression_type = _abi.EXPR_SUBJECT_TO, _abi.EXPR_MINIMIZE
    st.left_hand_regularization_cost_multiplier, st.right_hand_regularization_cost_multiplier = 0.7, 3.0
    st.right_point_position_expression_type = _abi.EXPR_SUBJECT_TO   # (rows BEHIND the hand rows: planner.py:385-399)
    st.com_position_expression_type = _abi.EXPR_SUBJECT_TO
    return st


def reference_settings(model, urdf_path, mine):
    s = pf.Settings()
    s.robot_urdf = urdf_path
    s.joints_name_list = list(model.joint_names)
    s.root_link = "root_link"
    s.desired_frame_quaternion_cost_frame_name = "chest"
    s.contact_points = hp_rp.FeetContactPointDescriptors()
    s.contact_points.left = hp_rp.ContactPointDescriptor.rectangular_foot("l_sole", 0.232, 0.1, np.array([0.116, 0.05, 0.0]))
    s.contact_points.right = hp_rp.ContactPointDescriptor.rectangular_foot("r_sole", 0.232, 0.1, np.array([0.116, 0.05, 0.0]))
    for k in ("relaxed_complementarity_epsilon", "static_friction", "maximum_joint_positions", "minimum_joint_positions",
              "joint_regularization_cost_weights", "base_quaternion_cost_multiplier", "desired_frame_quaternion_cost_multiplier",
              "joint_regularization_cost_multiplier", "force_regularization_cost_multiplier", "com_regularization_cost_multiplier",
              "average_force_regularization_cost_multiplier", "point_position_regularization_cost_multiplier",
              "lef_hand_position_in_frame", "right_hand_position_in_frame", "left_hand_regularization_cost_multiplier",
              "right_hand_regularization_cost_multiplier"):
        setattr(s, k, getattr(mine, k))
    expr = {_abi.EXPR_SKIP: hp.ExpressionType.skip, _abi.EXPR_SUBJECT_TO: hp.ExpressionType.subject_to, _abi.EXPR_MINIMIZE: hp.ExpressionType.minimize}
    for k in ("com_position_expression_type", "left_point_position_expression_type", "right_point_position_expression_type",
              "left_hand_expression_type", "right_hand_expression_type"):
        setattr(s, k, expr[getattr(mine, k)])
    s.left_hand_frame_name, s.right_hand_frame_name = "l_hand_palm", "r_hand_palm"
    s.casadi_function_options, s.casadi_opti_options, s.casadi_solver_options = {}, {}, {}
    return s


def main():
    model = synthetic_ergocub()
    mass = model.get_total_mass()
    mine = numeric_settings(model)

    class KinDyn:   # what Variables.__post_init__ reads of adam's KinDynComputations (planner.py:281-292)
        NDoF = model.NDoF
        get_total_mass = staticmethod(lambda: mass)

    with tempfile.TemporaryDirectory() as tmp:
        urdf_path = os.path.join(tmp, "synthetic_ergocub.urdf")
        open(urdf_path, "w").write(to_urdf(model, extra_frames=("l_hand_palm", "r_hand_palm")))
        settings = reference_settings(model, urdf_path, mine)
        assert settings.is_valid()
        variables = pf.Variables(settings=settings, kin_dyn_object=KinDyn())
        x, p = make_pose_workload(mine, model, 1, SEED)
        values, meta = variables.to_dicts()
        fill, xo, po = {}, 0, 0
        for name, val in values.items():
            size = int(np.asarray(val, dtype=float).size)
            if meta[name][hp.OptimizationObject.StorageTypeField] == "variable":
                arr = x[0, xo:xo + size].copy(); xo += size
            else:
                arr = p[0, po:po + size].copy(); po += size
            physical = name.endswith(".f") and ".contact_points." in name   # state and references (planner.py:788-850)
            fill[name] = (arr * mass if physical else arr).reshape(np.asarray(val, dtype=float).shape)
        assert xo == x.shape[1] and po == p.shape[1], (xo, po)
        variables.from_dict(fill)
        desc, xr, pr, numeric, model_r = pose_from_reference(settings, variables)
    blob = np.frombuffer(ctypes.string_at(ctypes.addressof(desc), ctypes.sizeof(desc)), dtype=np.uint8).copy()
    dst = os.path.join(ROOT, "tests", "golden", "pose_from_reference.npz")
    np.savez_compressed(dst, desc=blob, x=xr, p=pr, seed=SEED)
    print("x", xr.shape, "p", pr.shape, "desc bytes", blob.size, "max |x - workload|", np.abs(xr - x[0]).max(), "max |p - workload|", np.abs(pr - p[0]).max(), "->", dst)


if __name__ == "__main__":
    main()
