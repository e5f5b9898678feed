"""Numeric mirror of turnkey_planners/humanoid_pose_finder/planner.py:19-193 (Settings: the fields that reach the NLP), the
parameter packing of its Variables / References (:196-320) and synthetic pose workloads."""
import dataclasses

import numpy as np

from . import _abi
from .kinodyn_layout import rectangular_foot
from .kinodyn_settings import _fill_terrain_steps
from .robot_model import RobotModel

NJ = _abi.NJ

# parameter offsets (reference creation order, tests/golden/pose_*.npz "pnames")
P_DESC, P_MASS, P_LINK_LEN, P_LINK_DENS, P_GRAV, P_REF, P_REF_PB, P_REF_QB, P_REF_S, P_REF_COM = 0, 24, 25, 26, 27, 33, 105, 108, 112, 135
P_REF_FQ, P_REF_LH, P_REF_RH, P_EPS, P_MU, P_SMAX, P_SMIN, P_LH_IN, P_RH_IN = 138, 142, 145, 148, 149, 150, 173, 196, 199
# variable offsets
X_PB, X_QB, X_S, X_COM = 48, 51, 55, 78


@dataclasses.dataclass
class PoseSettings:
    """Defaults = Settings.__post_init__ (planner.py:76-91) + the constants of humanoid_pose_finder/main.py:57-101."""
    terrain: int = _abi.TERRAIN_PLANAR
    terrain_steps: list = dataclasses.field(default_factory=list)
    left_descriptors: np.ndarray = None
    right_descriptors: np.ndarray = None
    gravity: np.ndarray = dataclasses.field(default_factory=lambda: np.array([0.0, 0.0, -9.80665, 0.0, 0.0, 0.0]))
    relaxed_complementarity_epsilon: float = 0.0001
    static_friction: float = 0.3
    maximum_joint_positions: np.ndarray = None
    minimum_joint_positions: np.ndarray = None
    com_position_expression_type: int = _abi.EXPR_MINIMIZE
    left_point_position_expression_type: int = _abi.EXPR_MINIMIZE
    right_point_position_expression_type: int = _abi.EXPR_MINIMIZE
    base_quaternion_cost_multiplier: float = 50.0
    desired_frame_quaternion_cost_multiplier: float = 100.0
    com_regularization_cost_multiplier: float = 10.0
    joint_regularization_cost_weights: np.ndarray = None
    joint_regularization_cost_multiplier: float = 0.1
    force_regularization_cost_multiplier: float = 0.2
    average_force_regularization_cost_multiplier: float = 10.0
    point_position_regularization_cost_multiplier: float = 100.0
    # hand position expressions (planner.py:62-69, 86-91, 596-660).  The reference names the frames
    # (left_hand_frame_name / right_hand_frame_name); here a frame is (link index, link_R_frame [3,3], link_o_frame [3]) —
    # `hand_frame(model, link)` / `urdf_model.resolve_frame` produce it.  `lef_hand_position_in_frame` is the reference's spelling.
    left_hand_frame: tuple = None
    right_hand_frame: tuple = None
    lef_hand_position_in_frame: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(3))
    right_hand_position_in_frame: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(3))
    left_hand_regularization_cost_multiplier: float = 1.0
    right_hand_regularization_cost_multiplier: float = 1.0
    left_hand_expression_type: int = _abi.EXPR_SKIP
    right_hand_expression_type: int = _abi.EXPR_SKIP

    def __post_init__(self):
        if self.left_descriptors is None:
            self.left_descriptors = rectangular_foot(0.232, 0.1, [0.116, 0.05, 0.0])
        if self.right_descriptors is None:
            self.right_descriptors = rectangular_foot(0.232, 0.1, [0.116, 0.05, 0.0])
        if self.maximum_joint_positions is None:
            self.maximum_joint_positions = np.full(NJ, np.inf)
        if self.minimum_joint_positions is None:
            self.minimum_joint_positions = np.full(NJ, -np.inf)
        if self.joint_regularization_cost_weights is None:  # main.py:83-86
            w = np.ones(NJ)
            w[:3] = 0.1
            w[3:11] = 10.0
            self.joint_regularization_cost_weights = w

    def to_c(self) -> _abi.PoseSettingsC:
        s = _abi.PoseSettingsC()
        s.terrain = int(self.terrain)
        _fill_terrain_steps(s, self.terrain_steps)
        s.com_position_type = int(self.com_position_expression_type)
        s.left_point_position_type = int(self.left_point_position_expression_type)
        s.right_point_position_type = int(self.right_point_position_expression_type)
        s.base_quaternion_cost_multiplier = float(self.base_quaternion_cost_multiplier)
        s.desired_frame_quaternion_cost_multiplier = float(self.desired_frame_quaternion_cost_multiplier)
        s.com_regularization_cost_multiplier = float(self.com_regularization_cost_multiplier)
        for i in range(NJ):
            s.joint_regularization_cost_weights[i] = float(self.joint_regularization_cost_weights[i])
        s.joint_regularization_cost_multiplier = float(self.joint_regularization_cost_multiplier)
        s.force_regularization_cost_multiplier = float(self.force_regularization_cost_multiplier)
        s.average_force_regularization_cost_multiplier = float(self.average_force_regularization_cost_multiplier)
        s.point_position_regularization_cost_multiplier = float(self.point_position_regularization_cost_multiplier)
        hands = ((self.left_hand_expression_type, self.left_hand_frame, self.left_hand_regularization_cost_multiplier, "left"),
                 (self.right_hand_expression_type, self.right_hand_frame, self.right_hand_regularization_cost_multiplier, "right"))
        for h, (mode, frame, mult, side) in enumerate(hands):
            s.hand_type[h] = int(mode)
            s.hand_regularization_cost_multiplier[h] = float(mult)
            if int(mode) == _abi.EXPR_SKIP:
                continue
            if frame is None:   # planner.py:165-170, 180-185: "<side>_hand_frame_name is None"
                raise ValueError("%s_hand_frame is None but %s_hand_expression_type is not skip" % (side, side))
            link, R, o = frame
            s.hand_frame_link[h] = int(link)
            for i, v in enumerate(np.asarray(R, float).reshape(9)):
                s.hand_frame_R[h][i] = float(v)
            for i, v in enumerate(np.asarray(o, float).reshape(3)):
                s.hand_frame_o[h][i] = float(v)
        return s


def pose_finder_settings(model: RobotModel = None) -> PoseSettings:
    s = PoseSettings()
    if model is not None and model.max_joint_positions is not None:
        s.maximum_joint_positions = np.array(model.max_joint_positions, float)
        s.minimum_joint_positions = np.array(model.min_joint_positions, float)
    return s


def pack_pose_parameters(settings: PoseSettings, model: RobotModel, references: dict) -> np.ndarray:
    """p [202] in the reference's parameter creation order.  `references`: point_p [8,3], point_f [8,3] (already divided by
    the mass: planner.py:795-819), base_position, base_quaternion, joints, com, frame_quaternion."""
    p = np.zeros(_abi.POSE_NP)
    desc = np.concatenate([settings.left_descriptors, settings.right_descriptors]).reshape(8, 3)
    p[P_DESC:P_DESC + 24] = desc.reshape(-1)
    p[P_MASS] = model.get_total_mass()
    p[P_LINK_LEN] = 0.0   # non-parametric model: planner.py:266-270, :286-288
    p[P_LINK_DENS] = 0.0
    p[P_GRAV:P_GRAV + 6] = settings.gravity
    for c in range(8):
        p[P_REF + 9 * c:P_REF + 9 * c + 3] = references["point_p"][c]
        p[P_REF + 9 * c + 3:P_REF + 9 * c + 6] = references["point_f"][c]
        p[P_REF + 9 * c + 6:P_REF + 9 * c + 9] = desc[c]
    p[P_REF_PB:P_REF_PB + 3] = references["base_position"]
    p[P_REF_QB:P_REF_QB + 4] = references["base_quaternion"]
    p[P_REF_S:P_REF_S + NJ] = references["joints"]
    p[P_REF_COM:P_REF_COM + 3] = references["com"]
    p[P_REF_FQ:P_REF_FQ + 4] = references["frame_quaternion"]
    p[P_REF_LH:P_REF_LH + 3] = references.get("left_hand_position", np.zeros(3))      # planner.py:221-222: zeros by default
    p[P_REF_RH:P_REF_RH + 3] = references.get("right_hand_position", np.zeros(3))
    p[P_LH_IN:P_LH_IN + 3] = settings.lef_hand_position_in_frame
    p[P_RH_IN:P_RH_IN + 3] = settings.right_hand_position_in_frame
    p[P_EPS] = settings.relaxed_complementarity_epsilon
    p[P_MU] = settings.static_friction
    p[P_SMAX:P_SMAX + NJ] = settings.maximum_joint_positions
    p[P_SMIN:P_SMIN + NJ] = settings.minimum_joint_positions
    return p


def hand_frame(model: RobotModel, link: int, rpy=(0.0, 0.0, 0.0), offset=(0.0, 0.0, 0.0)) -> tuple:
    """A frame rigidly attached to `link` (what a URDF's fixed-joint hand frame resolves to): (link, link_R_frame, link_o_frame)."""
    from .robot_model import rot_from_rpy
    return int(link), rot_from_rpy(*rpy), np.asarray(offset, float)


def hand_point_position(settings: PoseSettings, model: RobotModel, hand: int, pb, quat_xyzw, s) -> np.ndarray:
    """World position of the hand point (numpy FK of the model): a reference for synthetic workloads and tests."""
    link, R, o = settings.left_hand_frame if hand == 0 else settings.right_hand_frame
    p_in = settings.lef_hand_position_in_frame if hand == 0 else settings.right_hand_position_in_frame
    Rl, ol = model.link_poses(pb, quat_xyzw, s)
    return ol[link] + Rl[link] @ (np.asarray(o, float) + np.asarray(R, float) @ np.asarray(p_in, float))


def make_pose_workload(settings: PoseSettings, model: RobotModel, batch: int = 1, seed: int = 3000):
    """Seeded near-feasible poses: joints random, contact points near the FK of the soles, forces sharing the weight."""
    rng = np.random.RandomState(seed)
    xs, ps = np.zeros((batch, _abi.POSE_NX)), np.zeros((batch, _abi.POSE_NP))
    desc = np.concatenate([settings.left_descriptors, settings.right_descriptors]).reshape(8, 3)
    for b in range(batch):
        s = np.clip(rng.uniform(-0.5, 0.5, NJ), np.maximum(settings.minimum_joint_positions, -3.0), np.minimum(settings.maximum_joint_positions, 3.0))
        pb = np.array([0.05 * rng.standard_normal(), 0.05 * rng.standard_normal(), 0.7]) + 0.01 * rng.standard_normal(3)
        q = np.array([0.0, 0.0, 0.0, 1.0]) + 0.05 * rng.standard_normal(4)
        qn = q / np.linalg.norm(q)
        frames = [model.frame_pose(f, pb, qn, s) for f in (_abi.FRAME_LEFT_SOLE, _abi.FRAME_RIGHT_SOLE)]
        x = xs[b]
        for c in range(8):
            R, o = frames[0 if c < 4 else 1]
            x[6 * c:6 * c + 3] = o + R @ desc[c] + 1e-3 * rng.standard_normal(3)
            fz = rng.uniform(0.0, 9.81 / 8)
            x[6 * c + 3:6 * c + 6] = [0.1 * fz * rng.standard_normal(), 0.1 * fz * rng.standard_normal(), fz]
        x[X_PB:X_PB + 3], x[X_QB:X_QB + 4], x[X_S:X_S + NJ] = pb, q, s
        x[X_COM:X_COM + 3] = model.com_position(pb, qn, s) + 1e-3 * rng.standard_normal(3)
        qr = np.array([0.0, 0.0, 0.0, 1.0]) + 0.1 * rng.standard_normal(4)
        fq = np.array([0.0, 0.0, 0.0, 1.0]) + 0.1 * rng.standard_normal(4)
        refs = {"point_p": x[:48].reshape(8, 6)[:, :3] + 0.02 * rng.standard_normal((8, 3)),
                "point_f": np.tile([0.0, 0.0, 9.81 / 8], (8, 1)) + 0.05 * rng.standard_normal((8, 3)),
                "base_position": pb + 0.01 * rng.standard_normal(3), "base_quaternion": qr / np.linalg.norm(qr),
                "joints": s + 0.1 * rng.standard_normal(NJ), "com": np.array([0.0, 0.0, 0.7]) + 0.02 * rng.standard_normal(3),
                "frame_quaternion": fq / np.linalg.norm(fq)}
        for hnd, (key, mode) in enumerate((("left_hand_position", settings.left_hand_expression_type), ("right_hand_position", settings.right_hand_expression_type))):
            if mode != _abi.EXPR_SKIP:   # a reference a few centimetres off where the hand is
                refs[key] = hand_point_position(settings, model, hnd, pb, qn, s) + 0.03 * rng.standard_normal(3)
        ps[b] = pack_pose_parameters(settings, model, refs)
    return xs, ps
