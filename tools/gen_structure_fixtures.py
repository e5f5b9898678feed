#!/usr/bin/env python3
"""Generate STRUCTURAL golden fixtures by importing the reference (hippopt) in THIS
container with the inert stubs under tools/refstub (casadi & co. are not installed).

What is exercised for real (reference code, unmodified, read from /root/reference/src):
  * hippopt.base.optimization_object.OptimizationObject._scan / to_dicts / to_list
    (base/optimization_object.py:64-319)
  * MultipleShootingSolver._extend_structure_to_horizon
    (base/multiple_shooting_solver.py:64-181)
  * the kinodynamic Settings / Variables dataclasses
    (turnkey_planners/humanoid_kinodynamic/variables.py, settings.py)
  * ContactPointDescriptor.rectangular_foot (robot_planning/variables/contacts.py:38-65)

What is NOT exercised: anything symbolic (CasADi), adam, liecasadi.

Output: tests/golden/kinodyn_structure.json  (names, sizes, variable/parameter tag, in the
reference's creation order = CasADi column order; final-state to_list order).
Run:  python3 tools/gen_structure_fixtures.py
The reference never travels to the GPU box; only the JSON does.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refstub"))
sys.path.insert(0, "/root/reference/src")

import numpy as np  # noqa: E402

import hippopt as hp  # noqa: E402
import hippopt.robot_planning as hp_rp  # noqa: E402
import hippopt.turnkey_planners.humanoid_kinodynamic.settings as ws  # noqa: E402
import hippopt.turnkey_planners.humanoid_kinodynamic.variables as wv  # noqa: E402
from hippopt.base.multiple_shooting_solver import MultipleShootingSolver  # noqa: E402

JOINTS = [
    "torso_pitch", "torso_roll", "torso_yaw",
    "l_shoulder_pitch", "l_shoulder_roll", "l_shoulder_yaw", "l_elbow",
    "r_shoulder_pitch", "r_shoulder_roll", "r_shoulder_yaw", "r_elbow",
    "l_hip_pitch", "l_hip_roll", "l_hip_yaw", "l_knee", "l_ankle_pitch", "l_ankle_roll",
    "r_hip_pitch", "r_hip_roll", "r_hip_yaw", "r_knee", "r_ankle_pitch", "r_ankle_roll",
]


class FakeKinDyn:
    """Only the three attributes Variables.__post_init__ reads (variables.py:319,333,353)."""

    NDoF = len(JOINTS)
    g = np.array([0.0, 0.0, -9.80665, 0.0, 0.0, 0.0])

    @staticmethod
    def get_total_mass():
        return 56.0


def make_settings():
    s = ws.Settings()
    s.robot_urdf = "<none>"
    s.joints_name_list = list(JOINTS)
    s.contact_points = hp_rp.FeetContactPointDescriptors()
    for side, frame in (("left", "l_sole"), ("right", "r_sole")):
        setattr(
            s.contact_points,
            side,
            hp_rp.ContactPointDescriptor.rectangular_foot(
                foot_frame=frame, x_length=0.232, y_length=0.1,
                top_left_point_position=np.array([0.116, 0.05, 0.0]),
            ),
        )
    s.horizon_length = 3
    s.time_step = 0.1
    nj = len(JOINTS)
    s.minimum_com_height = 0.3
    s.minimum_feet_lateral_distance = 0.1
    s.maximum_feet_relative_height = 0.05
    s.maximum_joint_positions = np.ones(nj)
    s.minimum_joint_positions = -np.ones(nj)
    s.maximum_joint_velocities = 2 * np.ones(nj)
    s.minimum_joint_velocities = -2 * np.ones(nj)
    return s


def main():
    settings = make_settings()
    variables = wv.Variables(settings=settings, kin_dyn_object=FakeKinDyn())
    out = {"joints": JOINTS, "horizons": {}}
    for horizon in (2, 3):
        ext = MultipleShootingSolver._extend_structure_to_horizon(variables, horizon=horizon)
        values, meta = ext.to_dicts()
        entries = []
        for name, val in values.items():
            arr = np.asarray(val, dtype=float)
            entries.append({
                "name": name,
                "size": int(arr.size),
                "type": meta[name][hp.OptimizationObject.StorageTypeField],
                "default": [float(v) for v in arr.flatten()],
            })
        out["horizons"][str(horizon)] = entries

    # final-state ordering used by planner.py:408-415  (to_list sorts the flat keys)
    sys_state = variables.system.to_humanoid_state()
    d = sys_state.to_dict()
    out["humanoid_state_to_list_order"] = [
        {"name": k, "size": int(np.asarray(d[k]).size)} for k in sorted(d.keys())
    ]
    d2 = variables.final_state.to_dict()
    out["final_state_to_list_order"] = [
        {"name": k, "size": int(np.asarray(d2[k]).size)} for k in sorted(d2.keys())
    ]
    # rectangular foot descriptor coordinates (contacts.py:38-65)
    out["left_descriptors"] = [
        [float(v) for v in np.asarray(p.position_in_foot_frame).flatten()]
        for p in settings.contact_points.left
    ]
    dst = os.path.join(HERE, "..", "tests", "golden", "kinodyn_structure.json")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    nvar = sum(e["size"] for e in out["horizons"]["3"] if e["type"] == "variable")
    npar = sum(e["size"] for e in out["horizons"]["3"] if e["type"] == "parameter")
    print("horizon 3: variables", nvar, "parameters", npar, "->", os.path.normpath(dst))


if __name__ == "__main__":
    main()
