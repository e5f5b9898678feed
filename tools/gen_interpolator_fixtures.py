#!/usr/bin/env python3
"""Golden vectors of the initial-guess interpolators by EXECUTING THE REFERENCE'S OWN
robot_planning/utilities/interpolators.py (linear_interpolator, quaternion_slerp, transform_interpolator,
foot_contact_state_interpolator, humanoid_state_interpolator) in this container, on numeric stand-ins of the few casadi /
liecasadi calls it makes on numbers (tools/numeric_standin; CasADi / liecasadi are not installed).  The scenario is the
guess generation of main_periodic_step.py:355-454 (contact phases of one step, two half-horizon interpolations).

Output: tests/golden/interpolators.npz.   Run:  python3 tools/gen_interpolator_fixtures.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(HERE, "refstub"), os.path.join(HERE, "numeric_standin"), "/root/reference/src", ROOT):
    sys.path.insert(0, p)
sys.path.insert(0, os.path.join(HERE, "numeric_standin"))

import numpy as np  # noqa: E402
import liecasadi  # noqa: E402  (numeric stand-in)

import hippopt.robot_planning as hp_rp  # noqa: E402


def scenario(seed=0):
    rng = np.random.RandomState(seed)
    horizon_length, dt = 30, 0.1
    horizon = horizon_length * dt
    step_length = 0.6
    desc = hp_rp.FeetContactPointDescriptors()
    desc.left = hp_rp.ContactPointDescriptor.rectangular_foot("l_sole", 0.232, 0.1, np.array([0.116, 0.05, 0.0]))
    desc.right = hp_rp.ContactPointDescriptor.rectangular_foot("r_sole", 0.232, 0.1, np.array([0.116, 0.05, 0.0]))
    yaw = 0.3
    qz = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
    phases = hp_rp.FeetContactPhasesDescriptor()
    phases.left = [
        hp_rp.FootContactPhaseDescriptor(transform=liecasadi.SE3.from_translation_and_rotation(np.array([0.0, 0.1, 0.0]), liecasadi.SO3.Identity()),
                                         mid_swing_transform=liecasadi.SE3.from_translation_and_rotation(np.array([step_length / 2, 0.1, 0.05]), liecasadi.SO3.Identity()),
                                         force=np.array([0, 0, 100.0]), activation_time=None, deactivation_time=horizon / 6.0),
        hp_rp.FootContactPhaseDescriptor(transform=liecasadi.SE3.from_translation_and_rotation(np.array([step_length, 0.1, 0.0]), liecasadi.SO3(qz)),
                                         mid_swing_transform=None, force=np.array([0, 0, 100.0]), activation_time=horizon / 3.0, deactivation_time=None),
    ]
    phases.right = [
        hp_rp.FootContactPhaseDescriptor(transform=liecasadi.SE3.from_translation_and_rotation(np.array([step_length / 2, -0.1, 0.0]), liecasadi.SO3.Identity()),
                                         mid_swing_transform=liecasadi.SE3.from_translation_and_rotation(np.array([step_length, -0.1, 0.05]), liecasadi.SO3.Identity()),
                                         force=np.array([0, 0, 100.0]), activation_time=None, deactivation_time=horizon * 2.0 / 3.0),
        hp_rp.FootContactPhaseDescriptor(transform=liecasadi.SE3.from_translation_and_rotation(np.array([1.5 * step_length, -0.1, 0.0]), liecasadi.SO3.Identity()),
                                         mid_swing_transform=None, force=np.array([0, 0, 100.0]), activation_time=horizon * 5.0 / 6.0, deactivation_time=None),
    ]

    def state(shift):
        s = hp_rp.HumanoidState(contact_point_descriptors=desc, number_of_joints=23)
        s.kinematics.base.position = np.array([shift, 0.0, 0.7]) + 0.01 * rng.standard_normal(3)
        q = np.array([0.0, 0.0, 0.0, 1.0]) + 0.2 * rng.standard_normal(4)
        s.kinematics.base.quaternion_xyzw = q / np.linalg.norm(q)
        s.kinematics.joints.positions = 0.3 * rng.standard_normal(23)
        s.com = np.array([shift, 0.0, 0.6]) + 0.01 * rng.standard_normal(3)
        return s
    return horizon_length, dt, desc, phases, state(0.0), state(0.3), state(0.6)


def pack(states):
    """[knots][48 + 7 + 23 + 3]: per point p, f ; base position, quaternion ; joints ; com"""
    rows = []
    for s in states:
        pts = s.contact_points.left + s.contact_points.right
        rows.append(np.concatenate([np.concatenate([np.asarray(pt.p, float).reshape(-1), np.asarray(pt.f, float).reshape(-1)]) for pt in pts]
                                   + [np.asarray(s.kinematics.base.position, float).reshape(-1), np.asarray(s.kinematics.base.quaternion_xyzw, float).reshape(-1),
                                      np.asarray(s.kinematics.joints.positions, float).reshape(-1), np.asarray(s.com, float).reshape(-1)]))
    return np.array(rows)


def main():
    N, dt, desc, phases, s0, s1, s2 = scenario()
    h1 = N // 2
    first = hp_rp.humanoid_state_interpolator(initial_state=s0, final_state=s1, contact_phases=phases, contact_descriptor=desc,
                                              number_of_points=h1, dt=dt)
    second = hp_rp.humanoid_state_interpolator(initial_state=s1, final_state=s2, contact_phases=phases, contact_descriptor=desc,
                                               number_of_points=N - h1, dt=dt, t0=h1 * dt)
    guess = pack(first + second)
    # a window that starts in the middle of a swing (the recursion of interpolators.py:236-247) and the single-phase shortcut
    mid = hp_rp.feet_contact_points_interpolator(phases=phases, descriptor=desc, number_of_points=7, dt=dt, t0=0.72)
    mid_pts = np.array([np.concatenate([np.concatenate([np.asarray(pt.p, float).reshape(-1), np.asarray(pt.f, float).reshape(-1)])
                                        for pt in (m.left + m.right)]) for m in mid])
    lin = np.array([np.asarray(v).reshape(-1) for v in hp_rp.linear_interpolator(np.array([0.0, 1.0, 2.0]), np.array([3.0, -1.0, 2.5]), 5)])
    qa = np.array([0.1, -0.2, 0.3, 0.9]); qa /= np.linalg.norm(qa)
    qb = np.array([-0.3, 0.1, 0.5, 0.7]); qb /= np.linalg.norm(qb)
    sl = np.array([np.asarray(v).reshape(-1) for v in hp_rp.quaternion_slerp(qa, qb, 6)])
    same = np.array([np.asarray(v).reshape(-1) for v in hp_rp.quaternion_slerp(qa, qa, 3)])   # angle < 1e-6 branch
    out = os.path.join(ROOT, "tests", "golden", "interpolators.npz")
    np.savez_compressed(out, guess=guess, mid_points=mid_pts, linear=lin, slerp=sl, slerp_same=same, qa=qa, qb=qb,
                        s0=pack([s0])[0], s1=pack([s1])[0], s2=pack([s2])[0], horizon=N, dt=dt)
    print("guess", guess.shape, "mid", mid_pts.shape, "->", os.path.relpath(out, ROOT))


if __name__ == "__main__":
    main()
