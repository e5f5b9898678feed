"""hippopt_amd.robot_planning.batched_guess: 16 contact-phase descriptions -> one [16][n] block of decision vectors, against the
one-at-a-time path (humanoid_state_interpolator, pinned to the reference's own functions by tests/golden/interpolators.npz) copied
leaf by leaf into x the way Planner.set_initial_guess does."""
import numpy as np
import pytest

import hippopt_amd.robot_planning as hp_rp
from hippopt_amd.robot_planning.batched_guess import batched_guess_block, foot_schedule
from hippopt_amd.robot_planning.transforms import SE3, SO3


def _quat(yaw, tilt=0.0):
    q = np.array([tilt * 0.3, -tilt * 0.2, np.sin(yaw / 2), np.cos(yaw / 2)])
    return q / np.linalg.norm(q)


def make_case(g, horizon_time, rng):
    """one guess: a walking phase list in the style of main_periodic_step.py:360-413 with its own step length, timing and yaw"""
    L = 0.3 + 0.05 * g
    yaw = 0.03 * g
    T = horizon_time
    ph = hp_rp.FeetContactPhasesDescriptor()
    ph.left = [
        hp_rp.FootContactPhaseDescriptor(transform=SE3.from_translation_and_rotation(np.array([0.0, 0.1, 0.0]), SO3(_quat(0.0))),
                                         mid_swing_transform=SE3.from_translation_and_rotation(np.array([0.5 * L, 0.1, 0.05]), SO3(_quat(yaw, 0.1))),
                                         force=np.array([0.0, 0.0, 100.0 + g]), deactivation_time=T / 6.0 + 0.013 * g),
        hp_rp.FootContactPhaseDescriptor(transform=SE3.from_translation_and_rotation(np.array([L, 0.1, 0.0]), SO3(_quat(yaw))),
                                         force=np.array([0.0, 0.0, 100.0]), activation_time=T / 3.0 + 0.02 * g)]
    ph.right = [
        hp_rp.FootContactPhaseDescriptor(transform=SE3.from_translation_and_rotation(np.array([0.5 * L, -0.1, 0.0]), SO3(_quat(0.0))),
                                         force=np.array([1.0, 0.0, 90.0]), deactivation_time=T * 2.0 / 3.0),   # default mid-swing
        hp_rp.FootContactPhaseDescriptor(transform=SE3.from_translation_and_rotation(np.array([1.5 * L, -0.1, 0.0]), SO3(_quat(-yaw))),
                                         force=np.array([0.0, 0.0, 100.0]), activation_time=T * 5.0 / 6.0)]
    desc = hp_rp.FeetContactPointDescriptors()
    desc.left = hp_rp.ContactPointDescriptor.rectangular_foot("l_sole", 0.232, 0.1, [0.116, 0.05, 0.0])
    desc.right = hp_rp.ContactPointDescriptor.rectangular_foot("r_sole", 0.232, 0.1, [0.116, 0.05, 0.0])

    def state(shift):
        s = hp_rp.HumanoidState(contact_point_descriptors=desc, number_of_joints=23)
        s.kinematics.base.position = np.array([shift, 0.0, 0.6]) + 0.01 * rng.standard_normal(3)
        s.kinematics.base.quaternion_xyzw = _quat(0.1 * shift + 0.02 * g, 0.05)
        s.kinematics.joints.positions = 0.3 * rng.standard_normal(23)
        s.com = np.array([shift, 0.0, 0.55]) + 0.01 * rng.standard_normal(3)
        return s
    return ph, desc, state(0.0), state(L)


def reference_block(init, fin, phases, desc, N, dt, mass, t0):
    x = np.zeros((len(phases), 189 * N + 6))
    for g in range(len(phases)):
        states = hp_rp.humanoid_state_interpolator(initial_state=init[g], final_state=fin[g], contact_phases=phases[g], contact_descriptor=desc,
                                                   number_of_points=N, dt=dt, t0=t0)
        for k, s in enumerate(states):
            r = x[g, 189 * k:189 * (k + 1)]
            for c, pt in enumerate(s.contact_points.left + s.contact_points.right):
                r[15 * c + 6:15 * c + 9] = np.asarray(pt.p, float).reshape(3)
                r[15 * c + 9:15 * c + 12] = np.asarray(pt.f, float).reshape(3) / mass
            r[127:130] = np.asarray(s.kinematics.base.position, float).reshape(3)
            r[130:134] = np.asarray(s.kinematics.base.quaternion_xyzw, float).reshape(4)
            r[157:180] = np.asarray(s.kinematics.joints.positions, float).reshape(23)
            r[180:183] = np.asarray(s.com, float).reshape(3)
    return x


@pytest.mark.parametrize("N,dt,t0", [(30, 0.1, 0.0), (50, 0.1, 0.0), (15, 0.1, 1.5), (200, 0.025, 0.0)])
def test_batched_block_equals_the_one_at_a_time_guesses(N, dt, t0):
    """t0 = 1.5 starts the window of the second half of a trajectory (main_periodic_step.py:445-454), inside or after swings"""
    rng = np.random.RandomState(N)
    B, mass = 16, 56.3
    T = 3.0 if t0 > 0 else N * dt
    cases = [make_case(g, T, rng) for g in range(B)]
    phases, desc = [c[0] for c in cases], cases[0][1]
    init, fin = [c[2] for c in cases], [c[3] for c in cases]
    x = batched_guess_block(init, fin, phases, desc, N, dt, mass, t0=t0)
    ref = reference_block(init, fin, phases, desc, N, dt, mass, t0)
    assert x.shape == (B, 189 * N + 6)
    assert np.max(np.abs(x - ref)) <= 1e-15
    assert np.count_nonzero(x[:, 9:12]) > 0 and np.all(x[:, 189 * N:] == 0.0)   # forces present; the global variables keep the default


def test_schedule_errors_match_the_reference_messages():
    ph = [hp_rp.FootContactPhaseDescriptor(activation_time=0.5)]
    with pytest.raises(ValueError, match="first phase activation time"):
        foot_schedule(ph, 10, 0.1)
    ph = [hp_rp.FootContactPhaseDescriptor(deactivation_time=0.3), hp_rp.FootContactPhaseDescriptor()]
    with pytest.raises(ValueError, match="Phase 1 has no activation time"):
        foot_schedule(ph, 10, 0.1)
    ph = [hp_rp.FootContactPhaseDescriptor(deactivation_time=0.6), hp_rp.FootContactPhaseDescriptor(activation_time=0.4)]
    with pytest.raises(ValueError, match="greater than the activation time of the next phase"):
        foot_schedule(ph, 10, 0.1)
    keys, a, b, tau, st, f = foot_schedule([hp_rp.FootContactPhaseDescriptor()], 7, 0.1)
    assert st.all() and np.all(a == 0) and np.all(f[:, 2] == 100)
