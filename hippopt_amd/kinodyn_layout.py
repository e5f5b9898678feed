"""Flat layout of the kinodynamic NLP: where each named leaf of the reference's `Variables`
dataclass tree (turnkey_planners/humanoid_kinodynamic/variables.py:256-374) lives in the
decision vector x and in the parameter vector p.

Order = creation order of opti.variable()/opti.parameter() in the reference
(base/opti_solver.py:303-310), pinned by tests/golden/kinodyn_structure.json.
"""
import numpy as np

from . import _abi

NXK, NPK, NXG, NPG, NJ, NC = _abi.NXK, _abi.NPK, _abi.NXG, _abi.NPG, _abi.NJ, _abi.NC

# per-knot decision variables: (field suffix, offset, size); contact point c adds 15*c
POINT_FIELDS = (("v", 0, 3), ("f_dot", 3, 3), ("p", 6, 3), ("f", 9, 3), ("u_v", 12, 3))
V, FD, P, F, U, PT = 0, 3, 6, 9, 12, 15
VB, QD, PB, QB, SD, S, COM, H = 120, 123, 127, 130, 134, 157, 180, 183
KNOT_FIELDS = (
    ("kinematics.base.linear_velocity", VB, 3),
    ("kinematics.base.quaternion_velocity_xyzw", QD, 4),
    ("kinematics.base.position", PB, 3),
    ("kinematics.base.quaternion_xyzw", QB, 4),
    ("kinematics.joints.velocities", SD, NJ),
    ("kinematics.joints.positions", S, NJ),
    ("com", COM, 3),
    ("centroidal_momentum", H, 6),
)


def point_name(c):
    return f"contact_points.{'left' if c < 4 else 'right'}[{c % 4}]"


def variable_names(horizon):
    """[(name, offset, size)] of every decision variable, in x order."""
    out = []
    for k in range(horizon):
        base = NXK * k
        for c in range(NC):
            for suf, off, size in POINT_FIELDS:
                out.append((f"system[{k}].{point_name(c)}.{suf}", base + PT * c + off, size))
        for name, off, size in KNOT_FIELDS:
            out.append((f"system[{k}].{name}", base + off, size))
    out.append(("initial_state.centroidal_momentum", NXK * horizon, NXG))
    return out


class ParamLayout:
    """Offsets inside p (reference parameter creation order)."""

    STATE = {"p": 0, "f": 3, "descriptor": 6, "pb": 72, "qb": 75, "s": 79, "com": 102}
    REF = {"alpha_left": 0, "yaw_left": 4, "alpha_right": 5, "yaw_right": 9, "swing_height": 10,
           "centroid_weights": 11, "centroid": 14, "com_velocity": 17, "frame_quaternion": 20,
           "base_quaternion": 24, "base_quaternion_velocity": 28, "joint_regularization": 32}

    def __init__(self, horizon):
        self.N = int(horizon)
        self.g0 = 24 * self.N
        self.mass = self.g0
        self.init = self.g0 + 3
        self.fin = self.init + 105
        self.sc = self.fin + 105
        self.dt = self.sc
        self.gravity = self.sc + 1
        self.kt = self.sc + 7
        self.kbs = self.sc + 8
        self.eps = self.sc + 9
        self.mu = self.sc + 10
        self.umax = self.sc + 11
        self.fdmax = self.sc + 14
        self.lmax = self.sc + 17
        self.hmin = self.sc + 18
        self.dmin = self.sc + 19
        self.hmax = self.sc + 20
        self.jpmax = self.sc + 21
        self.jpmin = self.sc + 44
        self.jvmax = self.sc + 67
        self.jvmin = self.sc + 90
        self.np = 79 * self.N + 326

    def desc(self, k, c):
        return 24 * k + 3 * c

    def ref(self, k):
        return self.sc + 113 + 55 * k

    def parameter_names(self):
        """[(name, offset, size)] in p order — compared against the golden structure fixture."""
        out = []
        for k in range(self.N):
            for c in range(NC):
                out.append((f"system[{k}].{point_name(c)}.descriptor.position_in_foot_frame", self.desc(k, c), 3))
        out += [("mass", self.mass, 1), ("parametric_link_length_multipliers", self.mass + 1, 1),
                ("parametric_link_densities", self.mass + 2, 1)]
        for st, base in (("initial_state", self.init), ("final_state", self.fin)):
            for c in range(NC):
                out.append((f"{st}.{point_name(c)}.p", base + 9 * c, 3))
                out.append((f"{st}.{point_name(c)}.f", base + 9 * c + 3, 3))
                out.append((f"{st}.{point_name(c)}.descriptor.position_in_foot_frame", base + 9 * c + 6, 3))
            out.append((f"{st}.kinematics.base.position", base + 72, 3))
            out.append((f"{st}.kinematics.base.quaternion_xyzw", base + 75, 4))
            out.append((f"{st}.kinematics.joints.positions", base + 79, NJ))
            out.append((f"{st}.com", base + 102, 3))
        for name, off, size in (
            ("dt", self.dt, 1), ("gravity", self.gravity, 6), ("planar_dcc_height_multiplier", self.kt, 1),
            ("dcc_gain", self.kbs, 1), ("dcc_epsilon", self.eps, 1), ("static_friction", self.mu, 1),
            ("maximum_velocity_control", self.umax, 3), ("maximum_force_derivative", self.fdmax, 3),
            ("maximum_angular_momentum", self.lmax, 1), ("minimum_com_height", self.hmin, 1),
            ("minimum_feet_lateral_distance", self.dmin, 1), ("maximum_feet_relative_height", self.hmax, 1),
            ("maximum_joint_positions", self.jpmax, NJ), ("minimum_joint_positions", self.jpmin, NJ),
            ("maximum_joint_velocities", self.jvmax, NJ), ("minimum_joint_velocities", self.jvmin, NJ),
        ):
            out.append((name, off, size))
        for k in range(self.N):
            r = self.ref(k)
            for i in range(4):
                out.append((f"references[{k}].feet.left.points[{i}].desired_force_ratio", r + i, 1))
            out.append((f"references[{k}].feet.left.yaw", r + 4, 1))
            for i in range(4):
                out.append((f"references[{k}].feet.right.points[{i}].desired_force_ratio", r + 5 + i, 1))
            out.append((f"references[{k}].feet.right.yaw", r + 9, 1))
            out.append((f"references[{k}].feet.desired_swing_height", r + 10, 1))
            out.append((f"references[{k}].contacts_centroid_cost_weights", r + 11, 3))
            out.append((f"references[{k}].contacts_centroid", r + 14, 3))
            out.append((f"references[{k}].com_linear_velocity", r + 17, 3))
            out.append((f"references[{k}].desired_frame_quaternion_xyzw", r + 20, 4))
            out.append((f"references[{k}].base_quaternion_xyzw", r + 24, 4))
            out.append((f"references[{k}].base_quaternion_xyzw_velocity", r + 28, 4))
            out.append((f"references[{k}].joint_regularization", r + 32, NJ))
        return out


def rectangular_foot(x_length, y_length, top_left):
    """ContactPointDescriptor.rectangular_foot  (robot_planning/variables/contacts.py:38-65)."""
    tl = np.asarray(top_left, float)
    return np.stack([tl, tl + [-x_length, 0.0, 0.0], tl + [-x_length, -y_length, 0.0], tl + [0.0, -y_length, 0.0]])


def yaw_corner_indices(descriptors):
    """bottom-right, top-right, top-left point indices of one foot (planner.py:773-828)."""
    d = np.asarray(descriptors, float)
    centroid = d.mean(axis=0)
    br = tr = tl = None
    brv = trv = tlv = 0.0
    for i in range(d.shape[0]):
        r = d[i] - centroid
        if r[1] < 0 and r[0] < 0 and (br is None or r[0] * r[1] > brv):
            brv, br = r[0] * r[1], i
        elif r[1] < 0 < r[0] and (tr is None or r[0] * r[1] > trv):
            trv, tr = r[0] * r[1], i
        elif r[1] > 0 and r[0] > 0 and (tl is None or r[0] * r[1] > tlv):
            tlv, tl = r[0] * r[1], i
    assert br is not None and tr is not None and tl is not None
    assert br != tr and tr != tl and tl != br
    return br, tr, tl
