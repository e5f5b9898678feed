"""Initial-guess interpolators (hippopt_amd.robot_planning.interpolators) against golden vectors produced by the reference's own
robot_planning/utilities/interpolators.py run on numeric stand-ins (tools/gen_interpolator_fixtures.py): the guess generation
of main_periodic_step.py:355-454 (contact phases of one step, two half-horizon windows), a window starting inside a swing,
plain linear interpolation and quaternion slerp."""
import os

import numpy as np
import pytest

import hippopt_amd.robot_planning as hp_rp

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "interpolators.npz"))


def scenario():
    horizon_length, dt = int(Z["horizon"]), float(Z["dt"])
    horizon = horizon_length * dt
    step_length = 0.6
    desc = hp_rp.FeetContactPointDescriptors()
    desc.left = hp_rp.ContactPointDescriptor.rectangular_foot("l_sole", 0.232, 0.1, np.array([0.116, 0.05, 0.0]))
    desc.right = hp_rp.ContactPointDescriptor.rectangular_foot("r_sole", 0.232, 0.1, np.array([0.116, 0.05, 0.0]))
    yaw = 0.3
    qz = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
    T, I = hp_rp.SE3.from_translation_and_rotation, hp_rp.SO3.Identity  # noqa: E741
    phases = hp_rp.FeetContactPhasesDescriptor()
    phases.left = [
        hp_rp.FootContactPhaseDescriptor(transform=T(np.array([0.0, 0.1, 0.0]), I()), mid_swing_transform=T(np.array([step_length / 2, 0.1, 0.05]), I()),
                                         force=np.array([0, 0, 100.0]), activation_time=None, deactivation_time=horizon / 6.0),
        hp_rp.FootContactPhaseDescriptor(transform=T(np.array([step_length, 0.1, 0.0]), hp_rp.SO3(qz)), mid_swing_transform=None,
                                         force=np.array([0, 0, 100.0]), activation_time=horizon / 3.0, deactivation_time=None)]
    phases.right = [
        hp_rp.FootContactPhaseDescriptor(transform=T(np.array([step_length / 2, -0.1, 0.0]), I()), mid_swing_transform=T(np.array([step_length, -0.1, 0.05]), I()),
                                         force=np.array([0, 0, 100.0]), activation_time=None, deactivation_time=horizon * 2.0 / 3.0),
        hp_rp.FootContactPhaseDescriptor(transform=T(np.array([1.5 * step_length, -0.1, 0.0]), I()), mid_swing_transform=None,
                                         force=np.array([0, 0, 100.0]), activation_time=horizon * 5.0 / 6.0, deactivation_time=None)]

    def state(row):
        s = hp_rp.HumanoidState(contact_point_descriptors=desc, number_of_joints=23)
        s.kinematics.base.position, s.kinematics.base.quaternion_xyzw = row[48:51].copy(), row[51:55].copy()
        s.kinematics.joints.positions, s.com = row[55:78].copy(), row[78:81].copy()
        return s
    return horizon_length, dt, desc, phases, state(Z["s0"]), state(Z["s1"]), state(Z["s2"])


def pack(states):
    rows = []
    for s in states:
        pts = s.contact_points.left + s.contact_points.right
        rows.append(np.concatenate([np.concatenate([np.asarray(pt.p, float).reshape(-1), np.asarray(pt.f, float).reshape(-1)]) for pt in pts]
                                   + [np.asarray(s.kinematics.base.position, float).reshape(-1), np.asarray(s.kinematics.base.quaternion_xyzw, float).reshape(-1),
                                      np.asarray(s.kinematics.joints.positions, float).reshape(-1), np.asarray(s.com, float).reshape(-1)]))
    return np.array(rows)


def test_humanoid_guess_matches_the_reference():
    N, dt, desc, phases, s0, s1, s2 = scenario()
    h1 = N // 2
    first = hp_rp.humanoid_state_interpolator(initial_state=s0, final_state=s1, contact_phases=phases, contact_descriptor=desc,
                                              number_of_points=h1, dt=dt)
    second = hp_rp.humanoid_state_interpolator(initial_state=s1, final_state=s2, contact_phases=phases, contact_descriptor=desc,
                                               number_of_points=N - h1, dt=dt, t0=h1 * dt)
    mine = pack(first + second)
    assert mine.shape == Z["guess"].shape
    assert np.max(np.abs(mine - Z["guess"])) < 1e-14
    # the swing really happens: a point of the left foot leaves the ground and carries no force while it does
    assert mine[:, 2].max() > 0.04 and np.all(mine[mine[:, 2] > 1e-6, 5] == 0.0)


def test_window_starting_inside_a_swing_and_basic_interpolators():
    N, dt, desc, phases, *_ = scenario()
    mid = hp_rp.feet_contact_points_interpolator(phases=phases, descriptor=desc, number_of_points=7, dt=dt, t0=0.72)
    pts = np.array([np.concatenate([np.concatenate([np.asarray(pt.p, float).reshape(-1), np.asarray(pt.f, float).reshape(-1)])
                                    for pt in (m.left + m.right)]) for m in mid])
    assert np.max(np.abs(pts - Z["mid_points"])) < 1e-14
    lin = np.array([np.asarray(v).reshape(-1) for v in hp_rp.linear_interpolator(np.array([0.0, 1.0, 2.0]), np.array([3.0, -1.0, 2.5]), 5)])
    assert np.max(np.abs(lin - Z["linear"])) < 1e-15
    sl = np.array([np.asarray(v).reshape(-1) for v in hp_rp.quaternion_slerp(Z["qa"], Z["qb"], 6)])
    assert np.max(np.abs(sl - Z["slerp"])) < 1e-15
    assert np.allclose(np.linalg.norm(sl, axis=1), 1.0) and np.allclose(sl[0], Z["qa"]) and np.allclose(sl[-1], Z["qb"])
    same = np.array([np.asarray(v).reshape(-1) for v in hp_rp.quaternion_slerp(Z["qa"], Z["qa"], 3)])
    assert np.array_equal(same, Z["slerp_same"])


def test_phase_validation_errors():
    N, dt, desc, phases, *_ = scenario()
    import copy
    bad = copy.deepcopy(phases.left)
    bad[1].activation_time = None
    with pytest.raises(ValueError, match="no activation time"):
        hp_rp.foot_contact_state_interpolator(bad, desc.left, 5, dt)
    bad = copy.deepcopy(phases.left)
    bad[0].deactivation_time = 2.0
    with pytest.raises(ValueError, match="greater than the activation time of the next phase"):
        hp_rp.foot_contact_state_interpolator(bad, desc.left, 5, dt)
    bad = copy.deepcopy(phases.left)
    bad[1].deactivation_time = 1.5
    with pytest.raises(ValueError, match="before the end time"):
        hp_rp.foot_contact_state_interpolator(bad, desc.left, 30, dt)
    with pytest.raises(ValueError, match="shape"):
        hp_rp.linear_interpolator(np.zeros(3), np.zeros(4), 3)
