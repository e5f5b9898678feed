"""Initial-guess generators with the call surface of robot_planning/utilities/interpolators.py (what main_periodic_step.py:433-454
calls to build a guess), written on top of the build's own schedule compiler.

Design (not the reference's): a foot's phase list is first COMPILED into a per-knot schedule by `batched_guess.foot_schedule` —
for every knot the pair of key poses it lies between, the blend parameter and the stance flag — and every function here is a
blend of arrays over that schedule (one weight vector per call, no per-sample bookkeeping).  With one guess this is the B = 1
case of `batched_guess.batched_guess_block`; the object-returning functions below exist because the planner mirrors
(`turnkey_planners/*/planner.py`) accept `HumanoidState` lists as guesses, like the reference's.

What pins the behaviour: tests/golden/interpolators.npz — outputs of the reference's own functions on numeric stand-ins
(tools/gen_interpolator_fixtures.py): sample counts per stance / swing, the half-way default of the mid-swing pose, the window
that opens inside a swing, slerp near identical quaternions.
"""
import numpy as np

from .batched_guess import foot_schedule, slerp
from .transforms import SE3
from .variables import (FeetContactPoints, FloatingBaseSystemState, FootContactState, FreeFloatingObjectState, HumanoidState,
                        KinematicTreeState)


def _weights(count):
    return np.linspace(0.0, 1.0, int(count))


def _column(value):
    value = np.asarray(value, dtype=float) if not isinstance(value, np.ndarray) else value
    return value.reshape(-1, 1) if value.ndim < 2 else value


def _blend(first, second, weights):
    """[(1 - w) first + w second for w in weights] on column vectors"""
    first, second = _column(first), _column(second)
    if first.shape != second.shape:
        raise ValueError(f"cannot interpolate between values of shape {first.shape} and shape {second.shape}")
    return [(1 - w) * first + w * second for w in weights]


def _blend_quaternions(first, second, weights):
    """spherical blend of two xyzw quaternions, one column per weight (the first quaternion when the two coincide)"""
    first, second = np.asarray(first, float).reshape(4), np.asarray(second, float).reshape(4)
    rows = slerp(np.broadcast_to(first, (len(weights), 4)), np.broadcast_to(second, (len(weights), 4)), np.asarray(weights, float))
    return [row.reshape(4, 1) for row in rows]


def linear_interpolator(initial, final, number_of_points: int) -> list:
    if isinstance(initial, list) or isinstance(final, list):
        raise TypeError("linear_interpolator blends two arrays; lists of values are interpolated element by element by the caller")
    return _blend(initial, final, _weights(number_of_points))


def quaternion_slerp(initial, final, number_of_points: int) -> list:
    if isinstance(initial, list) or isinstance(final, list):
        raise TypeError("quaternion_slerp blends two quaternions")
    return _blend_quaternions(initial, final, _weights(number_of_points))


def transform_interpolator(initial: SE3, final: SE3, number_of_points: int) -> list:
    w = _weights(number_of_points)
    places = _blend(initial.translation(), final.translation(), w)
    turns = _blend_quaternions(initial.rotation().as_quat().coeffs(), final.rotation().as_quat().coeffs(), w)
    return [SE3.from_position_quaternion(p, q) for p, q in zip(places, turns)]


def foot_contact_state_interpolator(phases: list, descriptor: list, number_of_points: int, dt: float, t0: float = 0.0) -> list:
    """One FootContactState per knot of the window [t0, t0 + number_of_points dt): the foot pose of the knot is the blend of its
    two key poses in the compiled schedule; the points carry the phase's force in stance and none in swing."""
    keys, a, b, tau, in_stance, force = foot_schedule(phases, number_of_points, dt, t0)
    key_pos = np.stack([k[0] for k in keys])
    key_quat = np.stack([k[1] for k in keys])
    w = tau[:, None]
    positions = (1 - w) * key_pos[a] + w * key_pos[b]
    quaternions = slerp(key_quat[a], key_quat[b], tau)
    states = []
    for k in range(int(number_of_points)):
        foot = FootContactState.from_parent_frame_transform(descriptor=descriptor, transform=SE3(positions[k], quaternions[k]))
        load = force[k].reshape(3, 1) if in_stance[k] else np.zeros((3, 1))
        for point in foot:
            point.f = load
        states.append(foot)
    return states


def feet_contact_points_interpolator(phases, descriptor, number_of_points: int, dt: float, t0: float = 0.0) -> list:
    per_foot = {side: foot_contact_state_interpolator(getattr(phases, side), getattr(descriptor, side), number_of_points, dt, t0)
                for side in ("left", "right")}
    feet = []
    for k in range(int(number_of_points)):
        both = FeetContactPoints()
        both.left, both.right = per_foot["left"][k], per_foot["right"][k]
        feet.append(both)
    return feet


def free_floating_object_state_interpolator(initial_state, final_state, number_of_points: int) -> list:
    w = _weights(number_of_points)
    return [FreeFloatingObjectState(position=p, quaternion_xyzw=q)
            for p, q in zip(_blend(initial_state.position, final_state.position, w),
                            _blend_quaternions(initial_state.quaternion_xyzw, final_state.quaternion_xyzw, w))]


def kinematic_tree_state_interpolator(initial_state, final_state, number_of_points: int) -> list:
    n0, n1 = np.size(initial_state.positions), np.size(final_state.positions)
    if n0 != n1:
        raise ValueError(f"the two joint configurations differ in size ({n0} and {n1} joints)")
    return [KinematicTreeState(positions=s) for s in _blend(initial_state.positions, final_state.positions, _weights(number_of_points))]


def floating_base_system_state_interpolator(initial_state, final_state, number_of_points: int) -> list:
    bases = free_floating_object_state_interpolator(initial_state.base, final_state.base, number_of_points)
    trees = kinematic_tree_state_interpolator(initial_state.joints, final_state.joints, number_of_points)
    systems = []
    for base, tree in zip(bases, trees):
        system = FloatingBaseSystemState()
        system.base, system.joints = base, tree
        systems.append(system)
    return systems


def humanoid_state_interpolator(initial_state, final_state, contact_phases, contact_descriptor, number_of_points: int, dt: float,
                                t0: float = 0.0) -> list:
    """The guess of main_periodic_step.py:433-454: contact points from the phase schedules of the two feet, base / joints / com
    blended between the two boundary states."""
    feet = feet_contact_points_interpolator(contact_phases, contact_descriptor, number_of_points, dt, t0)
    systems = floating_base_system_state_interpolator(initial_state.kinematics, final_state.kinematics, number_of_points)
    coms = _blend(initial_state.com, final_state.com, _weights(number_of_points))
    joints = int(np.size(initial_state.kinematics.joints.positions))
    states = []
    for contact_points, kinematics, com in zip(feet, systems, coms):
        state = HumanoidState(contact_point_descriptors=contact_descriptor, number_of_joints=joints)
        state.contact_points, state.kinematics, state.com = contact_points, kinematics, com
        states.append(state)
    return states
