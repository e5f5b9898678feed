"""Initial-guess interpolators: mirror of robot_planning/utilities/interpolators.py (linear :24-49, slerp :52-75, transform
:78-98, foot contact phases :101-288, feet :291-318, floating base / joints / humanoid :321-448) on numpy.
Same arguments, same point counts and phase arithmetic (ceil / round / min rules), same error messages; pinned by
tests/golden/interpolators.npz, which the reference's own functions produced (tools/gen_interpolator_fixtures.py)."""
import copy

import numpy as np

from .transforms import SE3, slerp_step
from .variables import (FeetContactPoints, FloatingBaseSystemState, FootContactState, FreeFloatingObjectState, HumanoidState,
                        KinematicTreeState)


def _as_col(v):
    if isinstance(v, np.ndarray) and v.ndim < 2:
        return np.expand_dims(v, axis=1)
    return v


def linear_interpolator(initial, final, number_of_points: int) -> list:
    assert not isinstance(initial, list) and not isinstance(final, list)
    initial, final = _as_col(initial), _as_col(final)
    if hasattr(initial, "shape") and hasattr(final, "shape") and initial.shape != final.shape:
        raise ValueError(f"Initial value has shape {initial.shape}, but final value has shape {final.shape}.")
    return [(1 - t_i) * initial + t_i * final for t_i in np.linspace(start=0.0, stop=1.0, num=number_of_points)]


def quaternion_slerp(initial, final, number_of_points: int) -> list:
    assert not isinstance(initial, list) and not isinstance(final, list)
    initial, final = _as_col(np.asarray(initial, float)), _as_col(np.asarray(final, float))
    with np.errstate(invalid="ignore"):   # a dot product of 1 + ulp gives NaN, as cs.acos does: NaN > 1e-6 is false, `initial` is returned
        angle = np.arccos(float((initial.T @ final).reshape(-1)[0]))
    out = []
    for t_i in np.linspace(start=0.0, stop=1.0, num=number_of_points):
        out.append(slerp_step(initial, final, t_i) if abs(angle) > 1e-6 else initial)
    return out


def transform_interpolator(initial: SE3, final: SE3, number_of_points: int) -> list:
    lin = linear_interpolator(initial.translation(), final.translation(), number_of_points)
    quat = quaternion_slerp(initial.rotation().as_quat().coeffs(), final.rotation().as_quat().coeffs(), number_of_points)
    return [SE3.from_position_quaternion(lin[i], quat[i]) for i in range(number_of_points)]


def foot_contact_state_interpolator(phases: list, descriptor: list, number_of_points: int, dt: float, t0: float = 0.0) -> list:
    assert len(phases) > 0
    assert number_of_points > 0
    assert dt > 0.0
    end_time = t0 + dt * number_of_points
    phases_copy = copy.deepcopy(phases)
    if phases_copy[0].activation_time is None:
        deactivation_time = phases_copy[0].deactivation_time if phases_copy[0].deactivation_time is not None else t0
        phases_copy[0].activation_time = min(deactivation_time, t0) - dt
    if phases_copy[0].activation_time > t0:
        raise ValueError(f"The first phase activation time ({phases_copy[0].activation_time}) is after the start time ({t0}).")
    for i, phase in enumerate(phases_copy):
        if phase.activation_time is None:
            raise ValueError(f"Phase {i} has no activation time, but is not the first phase.")
    last = len(phases_copy) - 1
    if phases_copy[last].deactivation_time is None:
        phases_copy[last].deactivation_time = max(end_time, phases_copy[last].activation_time) + dt
    if phases_copy[last].deactivation_time < end_time:
        raise ValueError(f"The Last phase deactivation time ({phases_copy[last].deactivation_time}) is before "
                         f"the end time ({end_time}, computed from the inputs).")
    for i, phase in enumerate(phases_copy):
        if phase.deactivation_time is None:
            raise ValueError(f"Phase {i} has no deactivation time, but is not the last phase.")
        if phase.activation_time > phase.deactivation_time:
            raise ValueError(f"Phase {i} has an activation time ({phase.activation_time}) "
                             f"greater than its deactivation time ({phase.deactivation_time}).")
        if i < last and phase.deactivation_time > phases_copy[i + 1].activation_time:
            raise ValueError(f"Phase {i} has a deactivation time ({phase.deactivation_time}) "
                             f"greater than the activation time of the next phase ({phases_copy[i + 1].activation_time}).")
    output = []

    def append_stance_phase(stance_phase, points: int) -> None:
        for _ in range(points):
            foot_state = FootContactState.from_parent_frame_transform(descriptor=descriptor, transform=stance_phase.transform)
            for point in foot_state:
                point.f = stance_phase.force
            output.append(foot_state)

    def append_swing_phase(start_phase, end_phase, points: int) -> None:
        full_swing_points = int(np.ceil((end_phase.activation_time - start_phase.deactivation_time) / dt))
        if start_phase.mid_swing_transform is None:
            start_phase.mid_swing_transform = SE3.from_translation_and_rotation(
                (start_phase.transform.translation() + end_phase.transform.translation()) / 2, end_phase.transform.rotation())
        mid_swing_points = min(round(full_swing_points / 2), points)
        for transform in transform_interpolator(start_phase.transform, start_phase.mid_swing_transform, mid_swing_points):
            foot_state = FootContactState.from_parent_frame_transform(descriptor=descriptor, transform=transform)
            for point in foot_state:
                point.f = np.zeros((3, 1))
            output.append(foot_state)
        second_half_points = points - mid_swing_points
        if second_half_points == 0:
            return
        for transform in transform_interpolator(start_phase.mid_swing_transform, end_phase.transform, second_half_points):
            foot_state = FootContactState.from_parent_frame_transform(descriptor=descriptor, transform=transform)
            for point in foot_state:
                point.f = np.zeros((3, 1))
            output.append(foot_state)

    if len(phases_copy) == 1 or phases_copy[0].deactivation_time >= end_time:
        append_stance_phase(phases_copy[0], number_of_points)
        return output
    i = 0
    activation_time = phases_copy[0].activation_time
    while activation_time < t0:
        if phases_copy[i].deactivation_time > t0:
            break
        i += 1
        activation_time = phases_copy[i].activation_time
    if activation_time > t0:   # the window starts inside a swing: start from the last stance and drop the advance
        previous_active_phase = phases_copy[i - 1]
        new_t0 = previous_active_phase.deactivation_time - dt
        advance_points = int(np.ceil((t0 - new_t0) / dt))
        increased_output = foot_contact_state_interpolator(phases=phases_copy, descriptor=descriptor,
                                                           number_of_points=number_of_points + advance_points, dt=dt, t0=new_t0)
        return increased_output[advance_points:]
    remaining_points = number_of_points
    while i < len(phases_copy) - 1:
        phase, next_phase = phases_copy[i], phases_copy[i + 1]
        stance_points = int(np.ceil((phase.deactivation_time - max(phase.activation_time, t0)) / dt))
        stance_points = min(stance_points, remaining_points)
        append_stance_phase(phase, stance_points)
        remaining_points -= stance_points
        if remaining_points == 0:
            return output
        swing_points = int(np.ceil((next_phase.activation_time - phase.deactivation_time) / dt))
        swing_points = min(swing_points, remaining_points)
        if swing_points == 0:
            continue
        append_swing_phase(phase, next_phase, swing_points)
        remaining_points -= swing_points
        if remaining_points == 0:
            return output
        i += 1
    append_stance_phase(phases_copy[len(phases_copy) - 1], remaining_points)
    return output


def feet_contact_points_interpolator(phases, descriptor, number_of_points: int, dt: float, t0: float = 0.0) -> list:
    left = foot_contact_state_interpolator(phases=phases.left, descriptor=descriptor.left, number_of_points=number_of_points, dt=dt, t0=t0)
    right = foot_contact_state_interpolator(phases=phases.right, descriptor=descriptor.right, number_of_points=number_of_points, dt=dt, t0=t0)
    assert len(left) == len(right) == number_of_points
    out = []
    for lft, rgt in zip(left, right):
        pts = FeetContactPoints()
        pts.left, pts.right = lft, rgt
        out.append(pts)
    return out


def free_floating_object_state_interpolator(initial_state, final_state, number_of_points: int) -> list:
    pos = linear_interpolator(initial_state.position, final_state.position, number_of_points)
    quat = quaternion_slerp(initial_state.quaternion_xyzw, final_state.quaternion_xyzw, number_of_points)
    assert len(pos) == len(quat) == number_of_points
    return [FreeFloatingObjectState(position=p, quaternion_xyzw=q) for p, q in zip(pos, quat)]


def kinematic_tree_state_interpolator(initial_state, final_state, number_of_points: int) -> list:
    if isinstance(initial_state.positions, np.ndarray) and isinstance(final_state.positions, np.ndarray) \
            and len(initial_state.positions) != len(final_state.positions):
        raise ValueError(f"Initial state has {len(initial_state.positions)} joints, but final state has {len(final_state.positions)} joints.")
    return [KinematicTreeState(positions=p) for p in linear_interpolator(initial_state.positions, final_state.positions, number_of_points)]


def floating_base_system_state_interpolator(initial_state, final_state, number_of_points: int) -> list:
    base = free_floating_object_state_interpolator(initial_state.base, final_state.base, number_of_points)
    joints = kinematic_tree_state_interpolator(initial_state.joints, final_state.joints, number_of_points)
    assert len(base) == len(joints) == number_of_points
    out = []
    for b, j in zip(base, joints):
        s = FloatingBaseSystemState()
        s.base, s.joints = b, j
        out.append(s)
    return out


def humanoid_state_interpolator(initial_state, final_state, contact_phases, contact_descriptor, number_of_points: int, dt: float,
                                t0: float = 0.0) -> list:
    contacts = feet_contact_points_interpolator(phases=contact_phases, descriptor=contact_descriptor, number_of_points=number_of_points, dt=dt, t0=t0)
    kinematics = floating_base_system_state_interpolator(initial_state.kinematics, final_state.kinematics, number_of_points)
    coms = linear_interpolator(initial_state.com, final_state.com, number_of_points)
    assert len(contacts) == len(kinematics) == len(coms) == number_of_points
    out = []
    for points, kin, com in zip(contacts, kinematics, coms):
        joints = initial_state.kinematics.joints.positions
        number_of_joints = joints.shape[0] * joints.shape[1] if hasattr(joints, "shape") and len(joints.shape) == 2 else len(joints)
        state = HumanoidState(contact_point_descriptors=contact_descriptor, number_of_joints=number_of_joints)
        state.contact_points, state.kinematics, state.com = points, kin, com
        out.append(state)
    return out
