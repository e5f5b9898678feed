#!/usr/bin/env python3
"""Run hippopt_amd.from_reference on the REFERENCE'S OWN objects, in this container, and commit what comes out.

Imports the reference (hippopt, read from /root/reference/src) with the inert stubs under tools/refstub (casadi & co. are not
installed) and builds, with the reference's own classes,
  * a `Settings` filled the way main_periodic_step.py:56-108 fills it (settings.py:12-147), robot_urdf = a URDF file of the
    synthetic robot (the ergoCub URDF is not in the image),
  * `Variables(settings, kin_dyn_object)` (variables.py:256-374), expanded over the horizon by the reference's own
    `MultipleShootingSolver._extend_structure_to_horizon` (base/multiple_shooting_solver.py:64-181), filled through the reference's own
    `OptimizationObject.from_dict` with a seeded trajectory in PHYSICAL units (forces and momenta multiplied by the mass),
then hands both to hippopt_amd.from_reference.from_reference and stores (hipnlp_desc bytes, x, p) in
tests/golden/from_reference_periodic_N4.npz.  The tests replay the fixture: the bytes must equal what the build's own settings /
model classes produce, x and p must equal the seeded trajectory (mass regularisation applied by the adapter), and on the GPU the
engine created from the stored bytes must agree with the oracle.  Nothing of the reference travels: only the arrays do.

Run:  python3 tools/gen_from_reference_fixture.py
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
import hippopt.turnkey_planners.humanoid_kinodynamic.settings as ws  # noqa: E402
import hippopt.turnkey_planners.humanoid_kinodynamic.variables as wv  # noqa: E402
from hippopt.base.multiple_shooting_solver import MultipleShootingSolver  # noqa: E402

from hippopt_amd.from_reference import from_reference  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402
from hippopt_amd.urdf_model import to_urdf  # noqa: E402

N, SEED = 4, 4242


def reference_settings(model, urdf_path):
    """main_periodic_step.py:17-108, with the synthetic robot's URDF and limits"""
    s = ws.Settings()
    s.robot_urdf = urdf_path
    s.joints_name_list = list(model.joint_names)
    nj = len(s.joints_name_list)
    s.root_link = "root_link"
    s.horizon_length = N
    s.time_step = 0.1
    s.contact_points = hp_rp.FeetContactPointDescriptors()
    s.contact_points.left = hp_rp.ContactPointDescriptor.rectangular_foot(
        foot_frame="l_sole", x_length=0.232, y_length=0.1, top_left_point_position=np.array([0.116, 0.05, 0.0]))
    s.contact_points.right = hp_rp.ContactPointDescriptor.rectangular_foot(
        foot_frame="r_sole", x_length=0.232, y_length=0.1, top_left_point_position=np.array([0.116, 0.05, 0.0]))
    s.planar_dcc_height_multiplier = 10.0
    s.dcc_gain = 40.0
    s.dcc_epsilon = 0.005
    s.static_friction = 0.3
    s.maximum_velocity_control = [2.0, 2.0, 5.0]
    s.maximum_force_derivative = [500.0, 500.0, 500.0]
    s.maximum_angular_momentum = 5.0
    s.minimum_com_height = 0.3
    s.minimum_feet_lateral_distance = 0.1
    s.maximum_feet_relative_height = 0.05
    s.maximum_joint_positions = np.array(model.max_joint_positions, float)   # (the script reads them from the URDF through idyntree)
    s.minimum_joint_positions = np.array(model.min_joint_positions, float)
    s.maximum_joint_velocities = np.ones(nj) * 2.0
    s.minimum_joint_velocities = np.ones(nj) * -2.0
    s.joint_regularization_cost_weights = np.ones(nj)
    s.joint_regularization_cost_weights[:3] = 0.1
    s.joint_regularization_cost_weights[3:11] = 10.0
    s.joint_regularization_cost_weights[11:] = 1.0
    s.contacts_centroid_cost_multiplier = 0.0
    s.com_linear_velocity_cost_weights = [10.0, 0.1, 1.0]
    s.com_linear_velocity_cost_multiplier = 1.0
    s.desired_frame_quaternion_cost_frame_name = "chest"
    s.desired_frame_quaternion_cost_multiplier = 200.0
    s.base_quaternion_cost_multiplier = 50.0
    s.base_quaternion_velocity_cost_multiplier = 0.001
    s.joint_regularization_cost_multiplier = 10.0
    s.force_regularization_cost_multiplier = 10.0
    s.foot_yaw_regularization_cost_multiplier = 2000.0
    s.swing_foot_height_cost_multiplier = 1000.0
    s.contact_velocity_control_cost_multiplier = 5.0
    s.contact_force_control_cost_multiplier = 0.0001
    s.final_state_expression_type = hp.ExpressionType.subject_to
    s.periodicity_expression_type = hp.ExpressionType.subject_to
    return s


def main():
    model = synthetic_ergocub()
    mass = model.get_total_mass()

    class KinDyn:   # the three attributes Variables.__post_init__ reads of adam's KinDynComputations (variables.py:319,333,353)
        NDoF = model.NDoF
        g = np.array([0.0, 0.0, -9.80665, 0.0, 0.0, 0.0])
        get_total_mass = staticmethod(lambda: mass)

    with tempfile.TemporaryDirectory() as tmp:
        urdf_path = os.path.join(tmp, "synthetic_ergocub.urdf")
        open(urdf_path, "w").write(to_urdf(model))
        settings = reference_settings(model, urdf_path)
        assert settings.is_valid()
        variables = wv.Variables(settings=settings, kin_dyn_object=KinDyn())
        expanded = MultipleShootingSolver._extend_structure_to_horizon(variables, horizon=N)
        # a seeded trajectory (already mass-normalised, as the engine's synthetic workloads are) -> physical units -> the reference tree
        st = periodic_step_settings(N, model)
        x, p = make_workload(st, model, batch=1, seed=SEED)
        values, meta = expanded.to_dicts()
        fill, xo, po = {}, 0, 0
        for name, val in values.items():
            size = int(np.asarray(val).size)
            if meta[name][hp.OptimizationObject.StorageTypeField] == "variable":
                arr = x[0, xo:xo + size].copy(); xo += size
            else:
                arr = p[0, po:po + size].copy(); po += size
            physical = (name.endswith(".f") and ".contact_points." in name and not name.startswith("references")) or \
                       (name.endswith("centroidal_momentum") and not name.startswith("references"))
            fill[name] = (arr * mass if physical else arr).reshape(np.asarray(val).shape)
        assert xo == x.shape[1] and po == p.shape[1]
        expanded.from_dict(fill)
        desc, xr, pr, numeric, model_r = from_reference(settings, expanded)
    blob = np.frombuffer(ctypes.string_at(ctypes.addressof(desc), ctypes.sizeof(desc)), dtype=np.uint8).copy()
    dst = os.path.join(ROOT, "tests", "golden", "from_reference_periodic_N4.npz")
    np.savez_compressed(dst, desc=blob, x=xr, p=pr, seed=SEED, horizon=N)
    print("x", xr.shape, "p", pr.shape, "desc bytes", blob.size, "max |x - workload|", np.abs(xr - x[0]).max(), "max |p - workload|", np.abs(pr - p[0]).max(), "->", dst)


if __name__ == "__main__":
    main()
