"""Synthetic kinodynamic trajectories and parameters: the inputs of every parity
test and of bench.py.  Pure numpy, deterministic per seed."""
import numpy as np

from . import _abi
from . import kinodyn_layout as L
from .kinodyn_settings import KinodynSettings
from .robot_model import RobotModel

NJ, NC, NXK = _abi.NJ, _abi.NC, _abi.NXK


def pack_parameters(settings: KinodynSettings, model: RobotModel, initial_state=None, final_state=None,
                    references=None) -> np.ndarray:
    """Flat parameter vector p in reference order.  `initial_state`/`final_state`: dicts with keys
    p[8,3], f[8,3] (already mass-normalised, planner.py:932-982), pb, qb, s, com.
    `references`: dict of per-knot arrays (see ParamLayout.REF) or None for the dataclass defaults
    (variables.py:97-120)."""
    N = settings.horizon_length
    pl = L.ParamLayout(N)
    p = np.zeros(pl.np)
    desc = np.concatenate([settings.left_descriptors, settings.right_descriptors], axis=0)  # [8,3]
    for k in range(N):
        p[pl.desc(k, 0):pl.desc(k, 0) + 24] = desc.reshape(-1)
    p[pl.mass] = model.get_total_mass()
    p[pl.mass + 1] = 0.0  # parametric_link_length_multipliers (non-parametric model: variables.py:335-339)
    p[pl.mass + 2] = 0.0
    for st, base in ((initial_state, pl.init), (final_state, pl.fin)):
        for c in range(NC):
            p[base + 9 * c + 6: base + 9 * c + 9] = desc[c]
        p[base + 75 + 3] = 1.0  # identity quaternion default (floating_base.py:23-25)
        if st is None:
            continue
        for c in range(NC):
            p[base + 9 * c: base + 9 * c + 3] = st["p"][c]
            p[base + 9 * c + 3: base + 9 * c + 6] = st["f"][c]
        p[base + 72: base + 75] = st["pb"]
        p[base + 75: base + 79] = st["qb"]
        p[base + 79: base + 79 + NJ] = st["s"]
        p[base + 102: base + 105] = st["com"]
    p[pl.dt] = settings.time_step
    p[pl.gravity: pl.gravity + 6] = settings.gravity
    p[pl.kt] = settings.planar_dcc_height_multiplier
    p[pl.kbs] = settings.dcc_gain
    p[pl.eps] = settings.dcc_epsilon
    p[pl.mu] = settings.static_friction
    p[pl.umax: pl.umax + 3] = settings.maximum_velocity_control
    p[pl.fdmax: pl.fdmax + 3] = settings.maximum_force_derivative
    p[pl.lmax] = settings.maximum_angular_momentum
    p[pl.hmin] = settings.minimum_com_height
    p[pl.dmin] = settings.minimum_feet_lateral_distance
    p[pl.hmax] = settings.maximum_feet_relative_height
    p[pl.jpmax: pl.jpmax + NJ] = settings.maximum_joint_positions
    p[pl.jpmin: pl.jpmin + NJ] = settings.minimum_joint_positions
    p[pl.jvmax: pl.jvmax + NJ] = settings.maximum_joint_velocities
    p[pl.jvmin: pl.jvmin + NJ] = settings.minimum_joint_velocities
    R = pl.REF
    for k in range(N):
        r = pl.ref(k)
        p[r + R["alpha_left"]: r + R["alpha_left"] + 4] = 0.25   # 1/number_of_points (variables.py:20-25)
        p[r + R["alpha_right"]: r + R["alpha_right"] + 4] = 0.25
        p[r + R["swing_height"]] = 0.02                           # variables.py:66
        p[r + R["frame_quaternion"] + 3] = 1.0
        p[r + R["base_quaternion"] + 3] = 1.0
        if references is not None:
            for key, val in references.items():
                v = np.asarray(val[k], float).reshape(-1)
                p[r + R[key]: r + R[key] + v.size] = v
    return p


def random_references(settings: KinodynSettings, rng) -> dict:
    """Reference signals with every entry non-trivial (so no cost term is accidentally inactive)."""
    N = settings.horizon_length
    quat = lambda: (lambda q: q / np.linalg.norm(q))(np.array([0, 0, 0, 1.0]) + 0.05 * rng.standard_normal(4))  # noqa: E731
    al = rng.uniform(0.15, 0.35, (N, 4))
    ar = rng.uniform(0.15, 0.35, (N, 4))
    return {
        "alpha_left": al / al.sum(axis=1, keepdims=True),
        "alpha_right": ar / ar.sum(axis=1, keepdims=True),
        "yaw_left": 0.1 * rng.standard_normal((N, 1)),
        "yaw_right": 0.1 * rng.standard_normal((N, 1)),
        "swing_height": rng.uniform(0.01, 0.05, (N, 1)),
        "centroid_weights": rng.uniform(0.5, 2.0, (N, 3)),
        "centroid": np.stack([0.1 * np.arange(N) * settings.time_step, np.zeros(N), np.zeros(N)], axis=1),
        "com_velocity": np.tile([0.1, 0.0, 0.0], (N, 1)) + 0.01 * rng.standard_normal((N, 3)),
        "frame_quaternion": np.stack([quat() for _ in range(N)]),
        "base_quaternion": np.stack([quat() for _ in range(N)]),
        "base_quaternion_velocity": 0.01 * rng.standard_normal((N, 4)),
        "joint_regularization": 0.1 * rng.standard_normal((N, NJ)),
    }


def random_trajectory(settings: KinodynSettings, model: RobotModel, seed: int):
    """One synthetic trajectory x [n] (SURVEY §8d distributions) and the state dict of its first knot."""
    rng = np.random.RandomState(seed)
    N = settings.horizon_length
    dt = settings.time_step
    g = -settings.gravity[2]
    x = np.zeros(NXK * N + _abi.NXG)
    desc = np.concatenate([settings.left_descriptors, settings.right_descriptors], axis=0)
    lo = np.maximum(settings.minimum_joint_positions, -0.5)
    hi = np.minimum(settings.maximum_joint_positions, 0.5)
    states = []
    for k in range(N):
        xs = x[NXK * k: NXK * (k + 1)]
        s = np.clip(rng.uniform(-0.5, 0.5, NJ), lo, hi)
        sd = 0.3 * rng.standard_normal(NJ)
        pb = np.array([0.1 * k * dt, 0.0, 0.7]) + 0.01 * rng.standard_normal(3)
        qb = np.array([0.0, 0.0, 0.0, 1.0]) + 0.05 * rng.standard_normal(4)
        qb /= np.linalg.norm(qb)
        qb *= 1.0 + 0.01 * rng.standard_normal()  # the NLP does not keep the quaternion exactly unit
        xs[L.VB:L.VB + 3] = 0.1 * rng.standard_normal(3)
        xs[L.QD:L.QD + 4] = 0.1 * rng.standard_normal(4)
        xs[L.PB:L.PB + 3] = pb
        xs[L.QB:L.QB + 4] = qb
        xs[L.SD:L.SD + NJ] = sd
        xs[L.S:L.S + NJ] = s
        stance = rng.uniform(0, 1, 2) > 0.3
        pts = np.zeros((NC, 3))
        frc = np.zeros((NC, 3))
        for c in range(NC):
            R, o = model.frame_pose(0 if c < 4 else 1, pb, qb, s)
            pts[c] = o + R @ desc[c] + 1e-3 * rng.standard_normal(3)
            fz = rng.uniform(0, g / 8) * (1.0 if stance[c // 4] else 0.05)
            frc[c] = [0.1 * fz * rng.standard_normal(), 0.1 * fz * rng.standard_normal(), fz]
            o_ = L.PT * c
            xs[o_ + L.V:o_ + L.V + 3] = 0.1 * rng.standard_normal(3)
            xs[o_ + L.FD:o_ + L.FD + 3] = 0.1 * rng.standard_normal(3)
            xs[o_ + L.P:o_ + L.P + 3] = pts[c]
            xs[o_ + L.F:o_ + L.F + 3] = frc[c]
            xs[o_ + L.U:o_ + L.U + 3] = 0.1 * rng.standard_normal(3)
        com = model.com_position(pb, qb, s) + 1e-3 * rng.standard_normal(3)
        xs[L.COM:L.COM + 3] = com
        xs[L.H:L.H + 6] = 0.05 * rng.standard_normal(6)
        states.append({"p": pts, "f": frc, "pb": pb, "qb": qb / np.linalg.norm(qb), "s": s, "com": com})
    x[NXK * N:] = 0.05 * rng.standard_normal(_abi.NXG)
    return x, states


def make_workload(settings: KinodynSettings, model: RobotModel, batch: int = 1, seed: int = 1000):
    """x [batch, n], p [batch, np]: `batch` trajectories = base trajectory + N(0, 0.02^2)."""
    rng = np.random.RandomState(seed)
    x0, states = random_trajectory(settings, model, seed)
    refs = random_references(settings, rng)
    # initial / final state parameters: near the first / last knot, so rows are near-feasible
    p0 = pack_parameters(settings, model, initial_state=states[0], final_state=states[-1], references=refs)
    xs = [x0]
    for b in range(1, batch):
        r = np.random.RandomState(2000 + b)
        xs.append(x0 + 0.02 * r.standard_normal(x0.shape))
    return np.stack(xs), np.tile(p0, (batch, 1))


def place_on_step_flanks(x: np.ndarray, settings: KinodynSettings, seed: int = 9) -> np.ndarray:
    """Move the contact points and the com of every knot of x[(batch,) n] onto the flanks of the smooth terrain steps, where
    exp(-g^(2 side)) actually varies (it is 0 or 1 to machine precision almost everywhere else), so that the terrain
    derivatives up to third order take part in a parity check."""
    rng = np.random.RandomState(seed)
    steps = settings.terrain_steps
    xs = x.reshape(-1, x.shape[-1])
    for row in xs:
        for k in range(settings.horizon_length):
            for c in range(8):
                st = steps[(k + c) % len(steps)]
                a, b = rng.uniform(0.95, 1.01), rng.uniform(0.0, 0.9)
                if c % 3 == 0:
                    a, b = b, rng.uniform(0.95, 1.01)
                sx, sy = rng.choice([-1.0, 1.0]), rng.choice([-1.0, 1.0])
                co, sn = np.cos(st.get("orientation", 0.0)), np.sin(st.get("orientation", 0.0))
                qx, qy = sx * 0.5 * st["length"] * a, sy * 0.5 * st["width"] * b
                o = 189 * k + 15 * c + 6
                row[o + 0] = st["position"][0] + co * qx - sn * qy
                row[o + 1] = st["position"][1] + sn * qx + co * qy
                row[o + 2] = 0.05 + 0.05 * rng.standard_normal()
            st = steps[k % len(steps)]
            row[189 * k + 180] = st["position"][0] - 0.5 * st["length"] * rng.uniform(0.96, 1.0)
            row[189 * k + 181] = st["position"][1] + 0.1 * rng.standard_normal()
    return x
