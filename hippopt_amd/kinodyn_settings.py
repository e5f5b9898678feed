"""Numeric mirror of turnkey_planners/humanoid_kinodynamic/settings.py (the fields that reach the
NLP) plus the constants of the reference's main scripts used as benchmark parameters."""
import dataclasses

import numpy as np

from . import _abi
from .kinodyn_layout import rectangular_foot, yaw_corner_indices

NJ = _abi.NJ


@dataclasses.dataclass
class KinodynSettings:
    horizon_length: int = 30
    time_step: float = 0.1
    gravity: np.ndarray = dataclasses.field(default_factory=lambda: np.array([0.0, 0.0, -9.80665, 0.0, 0.0, 0.0]))
    terrain: int = _abi.TERRAIN_PLANAR
    terrain_steps: list = dataclasses.field(default_factory=list)  # SMOOTH_STEPS: dicts(length, width, height, position, orientation, edge_sharpness, side_sharpness)
    left_descriptors: np.ndarray = None
    right_descriptors: np.ndarray = None
    # Opti parameters (settings.py:100-125 defaults, overridden by the main scripts)
    planar_dcc_height_multiplier: float = 10.0
    dcc_gain: float = 20.0
    dcc_epsilon: float = 0.05
    static_friction: float = 0.3
    maximum_velocity_control: np.ndarray = dataclasses.field(default_factory=lambda: np.array([2.0, 2.0, 5.0]))
    maximum_force_derivative: np.ndarray = dataclasses.field(default_factory=lambda: np.array([100.0, 100.0, 100.0]))
    maximum_angular_momentum: float = 10.0
    minimum_com_height: float = 0.3
    minimum_feet_lateral_distance: float = 0.1
    maximum_feet_relative_height: float = 0.05
    maximum_joint_positions: np.ndarray = None
    minimum_joint_positions: np.ndarray = None
    maximum_joint_velocities: np.ndarray = None
    minimum_joint_velocities: np.ndarray = None
    # graph constants
    final_state_expression_type: int = _abi.EXPR_SKIP
    final_state_expression_weight: float = 1.0
    periodicity_expression_type: int = _abi.EXPR_SKIP
    periodicity_expression_weight: float = 1.0
    contacts_centroid_cost_multiplier: float = 0.0
    com_linear_velocity_cost_weights: np.ndarray = dataclasses.field(default_factory=lambda: np.array([10.0, 0.1, 1.0]))
    com_linear_velocity_cost_multiplier: float = 1.0
    desired_frame_quaternion_cost_multiplier: float = 200.0
    base_quaternion_cost_multiplier: float = 50.0
    base_quaternion_velocity_cost_multiplier: float = 0.001
    joint_regularization_cost_weights: np.ndarray = None
    joint_regularization_cost_multiplier: float = 10.0
    force_regularization_cost_multiplier: float = 10.0
    foot_yaw_regularization_cost_multiplier: float = 2000.0
    swing_foot_height_cost_multiplier: float = 1000.0
    contact_velocity_control_cost_multiplier: float = 5.0
    contact_force_control_cost_multiplier: float = 0.0001
    joint_reg_as_coded: bool = True

    def __post_init__(self):
        if self.left_descriptors is None:  # main_periodic_step.py:58-69
            self.left_descriptors = rectangular_foot(0.232, 0.1, [0.116, 0.05, 0.0])
        if self.right_descriptors is None:
            self.right_descriptors = rectangular_foot(0.232, 0.1, [0.116, 0.05, 0.0])
        if self.maximum_joint_velocities is None:
            self.maximum_joint_velocities = 2.0 * np.ones(NJ)
        if self.minimum_joint_velocities is None:
            self.minimum_joint_velocities = -2.0 * np.ones(NJ)
        if self.maximum_joint_positions is None:
            self.maximum_joint_positions = np.full(NJ, np.inf)
        if self.minimum_joint_positions is None:
            self.minimum_joint_positions = np.full(NJ, -np.inf)
        if self.joint_regularization_cost_weights is None:  # main_periodic_step.py:89-92
            w = np.ones(NJ)
            w[:3] = 0.1
            w[3:11] = 10.0
            self.joint_regularization_cost_weights = w

    def is_valid(self):
        return self.horizon_length >= 2 and self.time_step > 0 and len(self.joint_regularization_cost_weights) == NJ

    def to_c(self) -> _abi.SettingsC:
        s = _abi.SettingsC()
        s.horizon = int(self.horizon_length)
        s.terrain = int(self.terrain)
        s.final_state_type = int(self.final_state_expression_type)
        s.periodicity_type = int(self.periodicity_expression_type)
        s.joint_reg_as_coded = 1 if self.joint_reg_as_coded else 0
        for foot, d in enumerate((self.left_descriptors, self.right_descriptors)):
            for i, v in enumerate(yaw_corner_indices(d)):
                s.yaw_corner[foot][i] = int(v)
        s.final_state_weight = float(self.final_state_expression_weight)
        s.periodicity_weight = float(self.periodicity_expression_weight)
        s.contacts_centroid_cost_multiplier = float(self.contacts_centroid_cost_multiplier)
        for i in range(3):
            s.com_linear_velocity_cost_weights[i] = float(self.com_linear_velocity_cost_weights[i])
        s.com_linear_velocity_cost_multiplier = float(self.com_linear_velocity_cost_multiplier)
        s.desired_frame_quaternion_cost_multiplier = float(self.desired_frame_quaternion_cost_multiplier)
        s.base_quaternion_cost_multiplier = float(self.base_quaternion_cost_multiplier)
        s.base_quaternion_velocity_cost_multiplier = float(self.base_quaternion_velocity_cost_multiplier)
        for i in range(NJ):
            s.joint_regularization_cost_weights[i] = float(self.joint_regularization_cost_weights[i])
        s.joint_regularization_cost_multiplier = float(self.joint_regularization_cost_multiplier)
        s.force_regularization_cost_multiplier = float(self.force_regularization_cost_multiplier)
        s.foot_yaw_regularization_cost_multiplier = float(self.foot_yaw_regularization_cost_multiplier)
        s.swing_foot_height_cost_multiplier = float(self.swing_foot_height_cost_multiplier)
        s.contact_velocity_control_cost_multiplier = float(self.contact_velocity_control_cost_multiplier)
        s.contact_force_control_cost_multiplier = float(self.contact_force_control_cost_multiplier)
        _fill_terrain_steps(s, self.terrain_steps)
        return s


def _fill_terrain_steps(s, terrain_steps):
    """SmoothTerrain.step arguments (smooth_terrain.py:266-336) -> hipnlp_terrain_step[]."""
    if len(terrain_steps) > _abi.MAX_TERRAIN_STEPS:
        raise ValueError("at most %d terrain steps" % _abi.MAX_TERRAIN_STEPS)
    s.n_terrain_steps = len(terrain_steps)
    for i, st in enumerate(terrain_steps):
        t = s.terrain_steps[i]
        t.length, t.width, t.height = float(st["length"]), float(st["width"]), float(st["height"])
        for j in range(3):
            t.position[j] = float(st.get("position", (0.0, 0.0, 0.0))[j])
        t.orientation = float(st.get("orientation", 0.0))
        t.edge_sharpness = int(st.get("edge_sharpness", 5))
        t.side_sharpness = int(st.get("side_sharpness", 10))
        normal = st.get("top_normal_direction", st.get("top_normal"))        # SmoothTerrain.step(top_normal_direction=...): None = the flat top
        for j in range(3):
            t.top_normal[j] = 0.0 if normal is None else float(normal[j])


def periodic_step_settings(horizon=30, model=None) -> KinodynSettings:
    """Constants of main_periodic_step.py:56-108 (final state and periodicity are constraints)."""
    s = KinodynSettings(
        horizon_length=horizon, time_step=0.1, dcc_gain=40.0, dcc_epsilon=0.005, static_friction=0.3,
        maximum_velocity_control=np.array([2.0, 2.0, 5.0]), maximum_force_derivative=np.array([500.0, 500.0, 500.0]),
        maximum_angular_momentum=5.0, minimum_com_height=0.3, minimum_feet_lateral_distance=0.1,
        maximum_feet_relative_height=0.05, contacts_centroid_cost_multiplier=0.0,
        desired_frame_quaternion_cost_multiplier=200.0, joint_regularization_cost_multiplier=10.0,
        final_state_expression_type=_abi.EXPR_SUBJECT_TO, periodicity_expression_type=_abi.EXPR_SUBJECT_TO,
    )
    if model is not None and model.max_joint_positions is not None:
        s.maximum_joint_positions = np.array(model.max_joint_positions, float)
        s.minimum_joint_positions = np.array(model.min_joint_positions, float)
    return s


def single_step_settings(horizon=30, model=None) -> KinodynSettings:
    """Constants of main_single_step_flat_ground.py:54-106 (final state / periodicity skipped)."""
    s = periodic_step_settings(horizon, model)
    s.contacts_centroid_cost_multiplier = 100.0
    s.desired_frame_quaternion_cost_multiplier = 90.0
    s.joint_regularization_cost_multiplier = 0.1
    s.final_state_expression_type = _abi.EXPR_SKIP
    s.periodicity_expression_type = _abi.EXPR_SKIP
    return s


def stairs_terrain_steps(length=0.45, width=0.8, height=0.1):
    """The two-step stairs of main_walking_on_stairs.py:18-28; the script calls get_terrain(length=step_length / 2 = 0.45,
    width=0.8, height=step_height = 0.1) (:397-403)."""
    return [
        {"length": 2 * length, "width": width, "height": height, "position": (1.5 * length, 0.0, 0.0)},
        {"length": 0.9 * length, "width": width, "height": height, "position": (2 * length, 0.0, 0.0)},
    ]


def ramp_terrain_steps(length=0.45, width=0.8, height=0.1, normal_direction=(-0.2, 0.0, 1.0)):
    """The ramp of main_walking_on_ramp.py:18-30: ONE step of twice the length whose top is the plane with the given normal; the script
    calls get_ramp(length=step_length / 2 = 0.45, width=0.8, height=step_height = 0.1, normal_direction=[-0.2, 0, 1]) (:397-409)."""
    return [{"length": 2 * length, "width": width, "height": height, "position": (1.5 * length, 0.0, 0.0),
             "top_normal_direction": tuple(float(v) for v in normal_direction)}]


def ramp_settings(horizon=50, model=None) -> KinodynSettings:
    """main_walking_on_ramp.py: the stairs script's constants on the ramp terrain (smooth step with a sloped top)."""
    s = stairs_settings(horizon, model)
    s.terrain_steps = ramp_terrain_steps()
    return s


def stairs_settings(horizon=50, model=None) -> KinodynSettings:
    """Constants of main_walking_on_stairs.py:70-148 that differ from the periodic step: smooth two-step terrain, final-state
    constraint, no periodicity (the remaining gains / weights are the periodic-step ones)."""
    s = periodic_step_settings(horizon, model)
    s.terrain = _abi.TERRAIN_SMOOTH_STEPS
    s.terrain_steps = stairs_terrain_steps()
    s.periodicity_expression_type = _abi.EXPR_SKIP
    return s
