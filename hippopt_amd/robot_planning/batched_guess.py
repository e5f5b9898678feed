"""Batched initial guesses: B contact-phase descriptions -> ONE [B][189 N + 6] block of decision vectors for `HipNlp(batch=B)`.

The reference builds a guess one object at a time: `humanoid_state_interpolator` (robot_planning/utilities/interpolators.py:321-448,
use at main_periodic_step.py:433-454) returns N `HumanoidState` objects which `Planner.set_initial_guess` copies leaf by leaf into
the `Variables` tree.  BASELINE config 5 evaluates 16 guesses at once (receding-horizon shape): here the same guesses are produced
as arrays.  Per guess and foot a SCHEDULE is compiled from the phase list — for every knot: the two key transforms it lies between,
the interpolation parameter and whether the foot is in stance — with exactly the reference's point-count arithmetic (ceil / round /
min rules, the advance when the window starts inside a swing, interpolators.py:101-288); everything numeric (translations, slerp of
the foot and base orientations, the four corner points of every foot, forces, joints, com) is then evaluated for all B x N knots in
vectorised numpy and written straight into the x block at the engine's per-knot offsets (include/hipnlp.h), mass-normalised like
`Planner.set_initial_guess` does (planner.py:932-982).  Checked entry for entry against the one-at-a-time path
(tests/test_batched_guess.py).
"""
import copy

import numpy as np

from .. import _abi

NXK, NC = _abi.NXK, _abi.NC
# per-knot offsets of the leaves a HumanoidState guess fills (tests/golden/kinodyn_structure.json; nlp_defs.h)
PT_STRIDE, OFF_P, OFF_F = 15, 6, 9
OFF_PB, OFF_QB, OFF_S, OFF_COM, OFF_H = 127, 130, 157, 180, 183


def foot_schedule(phases, number_of_points, dt, t0=0.0):
    """(keys, a, b, tau, stance, force): `keys` the list of key transforms (SE3) of the phase list; knot k of the window interpolates
    keys[a[k]] -> keys[b[k]] at tau[k]; stance[k] says whether the foot carries force[k] (else zero).  Same point counts and
    errors as foot_contact_state_interpolator (interpolators.py:101-288)."""
    assert len(phases) > 0 and number_of_points > 0 and dt > 0.0
    end_time = t0 + dt * number_of_points
    ph = copy.deepcopy(phases)
    if ph[0].activation_time is None:
        deact = ph[0].deactivation_time if ph[0].deactivation_time is not None else t0
        ph[0].activation_time = min(deact, t0) - dt
    if ph[0].activation_time > t0:
        raise ValueError(f"The first phase activation time ({ph[0].activation_time}) is after the start time ({t0}).")
    for i, phase in enumerate(ph):
        if phase.activation_time is None:
            raise ValueError(f"Phase {i} has no activation time, but is not the first phase.")
    last = len(ph) - 1
    if ph[last].deactivation_time is None:
        ph[last].deactivation_time = max(end_time, ph[last].activation_time) + dt
    if ph[last].deactivation_time < end_time:
        raise ValueError(f"The Last phase deactivation time ({ph[last].deactivation_time}) is before "
                         f"the end time ({end_time}, computed from the inputs).")
    for i, phase in enumerate(ph):
        if phase.deactivation_time is None:
            raise ValueError(f"Phase {i} has no deactivation time, but is not the last phase.")
        if phase.activation_time > phase.deactivation_time:
            raise ValueError(f"Phase {i} has an activation time ({phase.activation_time}) "
                             f"greater than its deactivation time ({phase.deactivation_time}).")
        if i < last and phase.deactivation_time > ph[i + 1].activation_time:
            raise ValueError(f"Phase {i} has a deactivation time ({phase.deactivation_time}) "
                             f"greater than the activation time of the next phase ({ph[i + 1].activation_time}).")
    # key transforms: 2 i = phase i, 2 i + 1 = its mid-swing transform (default: half way, orientation of the next phase)
    keys = []
    for i, phase in enumerate(ph):
        keys.append((np.asarray(phase.transform.translation(), float).reshape(3), np.asarray(phase.transform.rotation().as_quat().coeffs(), float).reshape(4)))
        if phase.mid_swing_transform is not None:
            m = phase.mid_swing_transform
            keys.append((np.asarray(m.translation(), float).reshape(3), np.asarray(m.rotation().as_quat().coeffs(), float).reshape(4)))
        elif i < last:
            nxt = ph[i + 1].transform
            keys.append(((keys[-1][0] + np.asarray(nxt.translation(), float).reshape(3)) / 2, np.asarray(nxt.rotation().as_quat().coeffs(), float).reshape(4)))
        else:
            keys.append(keys[-1])
    forces = [np.asarray(phase.force, float).reshape(3) for phase in ph]
    rows = []   # (a, b, tau, stance, force)

    def stance(i, points):
        rows.extend([(2 * i, 2 * i, 0.0, True, forces[i])] * points)

    def ramp(a, b, points):
        for t in np.linspace(0.0, 1.0, points):
            rows.append((a, b, float(t), False, np.zeros(3)))

    def swing(i, points):
        full = int(np.ceil((ph[i + 1].activation_time - ph[i].deactivation_time) / dt))
        first = min(round(full / 2), points)
        ramp(2 * i, 2 * i + 1, first)
        if points - first > 0:
            ramp(2 * i + 1, 2 * (i + 1), points - first)

    def run(points, start):
        if len(ph) == 1 or ph[0].deactivation_time >= start + dt * points:
            stance(0, points)
            return
        i, act = 0, ph[0].activation_time
        while act < start:
            if ph[i].deactivation_time > start:
                break
            i += 1
            act = ph[i].activation_time
        if act > start:   # the window starts inside a swing: start from the end of the previous stance, drop the advance
            new_start = ph[i - 1].deactivation_time - dt
            advance = int(np.ceil((start - new_start) / dt))
            run(points + advance, new_start)
            del rows[:advance]
            return
        remaining = points
        while i < len(ph) - 1:
            n_st = min(int(np.ceil((ph[i].deactivation_time - max(ph[i].activation_time, start)) / dt)), remaining)
            stance(i, n_st)
            remaining -= n_st
            if remaining == 0:
                return
            n_sw = min(int(np.ceil((ph[i + 1].activation_time - ph[i].deactivation_time) / dt)), remaining)
            if n_sw == 0:
                continue
            swing(i, n_sw)
            remaining -= n_sw
            if remaining == 0:
                return
            i += 1
        stance(len(ph) - 1, remaining)

    run(number_of_points, t0)
    assert len(rows) == number_of_points
    a = np.array([r[0] for r in rows]); b = np.array([r[1] for r in rows])
    tau = np.array([r[2] for r in rows]); st = np.array([r[3] for r in rows])
    force = np.stack([r[4] for r in rows])
    return keys, a, b, tau, st, force


def slerp(q1, q2, t):
    """rows of q1 -> q2 at t (arrays [..., 4], [...]): quaternion_slerp of the reference (interpolators.py:52-75): the first
    quaternion unchanged when the angle between the two is below 1e-6, liecasadi's slerp_step otherwise"""
    dot = np.sum(q1 * q2, axis=-1)
    with np.errstate(invalid="ignore"):
        angle = np.arccos(dot)
    small = ~(np.abs(angle) > 1e-6)   # (also when rounding pushed the dot product past 1: the reference's `abs(angle) > 1e-6` is then false)
    safe = np.where(small, 1.0, angle)
    out = (np.sin((1.0 - t) * safe)[..., None] * q1 + np.sin(t * safe)[..., None] * q2) / np.sin(safe)[..., None]
    return np.where(small[..., None], q1, out)


def rotate(q, v):
    """R(q) v for xyzw quaternions [..., 4] (as liecasadi: no normalisation) and vectors [..., 3]"""
    u, w = q[..., :3], q[..., 3:4]
    c1 = np.cross(u, v)
    return v + 2.0 * w * c1 + 2.0 * np.cross(u, c1)


def batched_guess_block(initial_states, final_states, contact_phases, contact_descriptor, number_of_points, dt, mass, t0=0.0,
                        mass_regularization=True):
    """x [B][189 N + 6] for B guesses: initial_states / final_states: lists of HumanoidState (kinematics, com);
    contact_phases: list of FeetContactPhasesDescriptor; contact_descriptor: FeetContactPointDescriptors (shared).
    Every leaf a HumanoidState carries is filled (contact point p and f, base position / quaternion, joint positions, com);
    velocities, force derivatives, u_v and the momenta keep the dataclass defaults (zero), as with the reference's guess."""
    B, N = len(contact_phases), int(number_of_points)
    assert len(initial_states) == len(final_states) == B
    x = np.zeros((B, NXK * N + _abi.NXG))
    knots = x[:, :NXK * N].reshape(B, N, NXK)
    desc = [np.stack([np.asarray(d.position_in_foot_frame, float).reshape(3) for d in side])
            for side in (contact_descriptor.left, contact_descriptor.right)]
    if desc[0].shape[0] + desc[1].shape[0] != NC:
        raise ValueError("the engine is built for four contact points per foot")
    scale = 1.0 / mass if mass_regularization else 1.0
    for foot in range(2):
        A_t, A_q, B_t, B_q = (np.zeros((B, N, 3)), np.zeros((B, N, 4)), np.zeros((B, N, 3)), np.zeros((B, N, 4)))
        tau, force = np.zeros((B, N)), np.zeros((B, N, 3))
        for g in range(B):   # the integer bookkeeping of the phase lists: scalar, cheap; everything below is array arithmetic
            keys, a, b, t, st, f = foot_schedule(contact_phases[g].left if foot == 0 else contact_phases[g].right, N, dt, t0)
            kt, kq = np.stack([k[0] for k in keys]), np.stack([k[1] for k in keys])
            A_t[g], A_q[g], B_t[g], B_q[g], tau[g], force[g] = kt[a], kq[a], kt[b], kq[b], t, f
        pos = (1.0 - tau)[..., None] * A_t + tau[..., None] * B_t
        quat = slerp(A_q, B_q, tau)
        for c in range(desc[foot].shape[0]):
            base = PT_STRIDE * (4 * foot + c)
            knots[:, :, base + OFF_P:base + OFF_P + 3] = pos + rotate(quat, np.broadcast_to(desc[foot][c], pos.shape))
            knots[:, :, base + OFF_F:base + OFF_F + 3] = force * scale
    lin = np.linspace(0.0, 1.0, N)

    def leaf(states, get, n):
        return np.stack([np.asarray(get(s), float).reshape(n) for s in states])
    for off, n, get in ((OFF_PB, 3, lambda s: s.kinematics.base.position), (OFF_S, _abi.NJ, lambda s: s.kinematics.joints.positions),
                        (OFF_COM, 3, lambda s: s.com)):
        v0, v1 = leaf(initial_states, get, n), leaf(final_states, get, n)
        knots[:, :, off:off + n] = (1.0 - lin)[None, :, None] * v0[:, None, :] + lin[None, :, None] * v1[:, None, :]
    q0 = leaf(initial_states, lambda s: s.kinematics.base.quaternion_xyzw, 4)
    q1 = leaf(final_states, lambda s: s.kinematics.base.quaternion_xyzw, 4)
    knots[:, :, OFF_QB:OFF_QB + 4] = slerp(np.broadcast_to(q0[:, None, :], (B, N, 4)), np.broadcast_to(q1[:, None, :], (B, N, 4)),
                                           np.broadcast_to(lin[None, :], (B, N)))
    return x
