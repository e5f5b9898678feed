"""The `Variables` dataclass tree of the kinodynamic planner: 189 decision variables and 79 parameters per knot, 6 global
variables, 326 global parameters.  Mirror of turnkey_planners/humanoid_kinodynamic/variables.py:13-374 (same field names,
order and storage types; the flat order is pinned by tests/golden/kinodyn_structure.json)."""
import dataclasses

import numpy as np

from ... import robot_planning as hp_rp
from ...base import (CompositeType, OptimizationObject, Parameter, StorageType, Variable, default_composite_field,
                     default_storage_field)


@dataclasses.dataclass
class ContactReferences(OptimizationObject):
    desired_force_ratio: StorageType = default_storage_field(Parameter)
    number_of_points: dataclasses.InitVar[int] = dataclasses.field(default=None)

    def __post_init__(self, number_of_points):
        self.desired_force_ratio = 1.0 / number_of_points if number_of_points is not None and number_of_points > 0 else 0.0


@dataclasses.dataclass
class FootReferences(OptimizationObject):
    points: CompositeType = default_composite_field(factory=list, time_varying=False)
    yaw: StorageType = default_storage_field(Parameter)
    number_of_points: dataclasses.InitVar[int] = dataclasses.field(default=0)

    def __post_init__(self, number_of_points):
        number_of_points = number_of_points if number_of_points is not None else 0
        self.points = [ContactReferences(number_of_points=number_of_points) for _ in range(number_of_points)]
        self.yaw = 0.0


@dataclasses.dataclass
class FeetReferences(OptimizationObject):
    left: CompositeType = default_composite_field(factory=FootReferences, time_varying=False)
    right: CompositeType = default_composite_field(factory=FootReferences, time_varying=False)
    desired_swing_height: StorageType = default_storage_field(Parameter)
    number_of_points_left: dataclasses.InitVar[int] = dataclasses.field(default=0)
    number_of_points_right: dataclasses.InitVar[int] = dataclasses.field(default=0)

    def __post_init__(self, number_of_points_left, number_of_points_right):
        self.left = FootReferences(number_of_points=number_of_points_left)
        self.right = FootReferences(number_of_points=number_of_points_right)
        self.desired_swing_height = 0.02


@dataclasses.dataclass
class References(OptimizationObject):
    feet: CompositeType = default_composite_field(factory=FeetReferences, time_varying=False)
    contacts_centroid_cost_weights: StorageType = default_storage_field(Parameter)
    contacts_centroid: StorageType = default_storage_field(Parameter)
    com_linear_velocity: StorageType = default_storage_field(Parameter)
    desired_frame_quaternion_xyzw: StorageType = default_storage_field(Parameter)
    base_quaternion_xyzw: StorageType = default_storage_field(Parameter)
    base_quaternion_xyzw_velocity: StorageType = default_storage_field(Parameter)
    joint_regularization: StorageType = default_storage_field(Parameter)
    number_of_joints: dataclasses.InitVar[int] = dataclasses.field(default=0)
    number_of_points_left: dataclasses.InitVar[int] = dataclasses.field(default=0)
    number_of_points_right: dataclasses.InitVar[int] = dataclasses.field(default=0)

    def __post_init__(self, number_of_joints, number_of_points_left, number_of_points_right):
        self.feet = FeetReferences(number_of_points_left=number_of_points_left, number_of_points_right=number_of_points_right)
        self.contacts_centroid_cost_weights = np.zeros((3, 1))
        self.contacts_centroid = np.zeros((3, 1))
        self.com_linear_velocity = np.zeros((3, 1))
        self.desired_frame_quaternion_xyzw = np.array([[0.0], [0.0], [0.0], [1.0]])
        self.base_quaternion_xyzw = np.array([[0.0], [0.0], [0.0], [1.0]])
        self.base_quaternion_xyzw_velocity = np.zeros((4, 1))
        self.joint_regularization = np.zeros((number_of_joints, 1))


@dataclasses.dataclass
class ExtendedContactPoint(hp_rp.ContactPointState, hp_rp.ContactPointStateDerivative):
    u_v: StorageType = default_storage_field(Variable)

    def __post_init__(self, input_descriptor):
        hp_rp.ContactPointState.__post_init__(self, input_descriptor)
        hp_rp.ContactPointStateDerivative.__post_init__(self)
        self.u_v = np.zeros(3)

    def to_contact_point_state(self):
        out = hp_rp.ContactPointState()
        out.p, out.f, out.descriptor = self.p, self.f, self.descriptor
        return out


@dataclasses.dataclass
class FeetContactPointsExtended(OptimizationObject):
    left: list = default_composite_field(factory=list)
    right: list = default_composite_field(factory=list)

    def to_feet_contact_points(self):
        out = hp_rp.FeetContactPoints()
        out.left = hp_rp.FootContactState.from_list([pt.to_contact_point_state() for pt in self.left])
        out.right = hp_rp.FootContactState.from_list([pt.to_contact_point_state() for pt in self.right])
        return out


@dataclasses.dataclass
class ExtendedHumanoid(OptimizationObject):
    contact_points: CompositeType = default_composite_field(factory=FeetContactPointsExtended)
    kinematics: CompositeType = default_composite_field(cls=Variable, factory=hp_rp.FloatingBaseSystem)
    com: StorageType = default_storage_field(Variable)
    centroidal_momentum: StorageType = default_storage_field(Variable)
    contact_point_descriptors: dataclasses.InitVar[hp_rp.FeetContactPointDescriptors] = dataclasses.field(default=None)
    number_of_joints: dataclasses.InitVar[int] = dataclasses.field(default=None)

    def __post_init__(self, contact_point_descriptors, number_of_joints):
        if contact_point_descriptors is not None:
            self.contact_points.left = [ExtendedContactPoint(input_descriptor=pt) for pt in contact_point_descriptors.left]
            self.contact_points.right = [ExtendedContactPoint(input_descriptor=pt) for pt in contact_point_descriptors.right]
        self.com = np.zeros(3)
        self.centroidal_momentum = np.zeros(6)
        self.kinematics = hp_rp.FloatingBaseSystem(number_of_joints=number_of_joints)

    def to_humanoid_state(self):
        out = hp_rp.HumanoidState()
        out.kinematics = self.kinematics.to_floating_base_system_state()
        out.contact_points = self.contact_points.to_feet_contact_points()
        out.com = self.com
        return out


@dataclasses.dataclass
class ExtendedHumanoidState(hp_rp.HumanoidState):
    centroidal_momentum: StorageType = default_storage_field(Variable)

    def __post_init__(self, contact_point_descriptors, number_of_joints):
        hp_rp.HumanoidState.__post_init__(self, contact_point_descriptors=contact_point_descriptors, number_of_joints=number_of_joints)
        self.centroidal_momentum = np.zeros(6)


@dataclasses.dataclass
class Variables(OptimizationObject):
    system: CompositeType = default_composite_field(cls=Variable, factory=ExtendedHumanoid)
    mass: StorageType = default_storage_field(Parameter)
    parametric_link_length_multipliers: StorageType = default_storage_field(Parameter)
    parametric_link_densities: StorageType = default_storage_field(Parameter)
    initial_state: CompositeType = default_composite_field(cls=Parameter, factory=ExtendedHumanoidState, time_varying=False)
    final_state: CompositeType = default_composite_field(cls=Parameter, factory=hp_rp.HumanoidState, time_varying=False)
    dt: StorageType = default_storage_field(Parameter)
    gravity: StorageType = default_storage_field(Parameter)
    planar_dcc_height_multiplier: StorageType = default_storage_field(Parameter)
    dcc_gain: StorageType = default_storage_field(Parameter)
    dcc_epsilon: StorageType = default_storage_field(Parameter)
    static_friction: StorageType = default_storage_field(Parameter)
    maximum_velocity_control: StorageType = default_storage_field(Parameter)
    maximum_force_derivative: StorageType = default_storage_field(Parameter)
    maximum_angular_momentum: StorageType = default_storage_field(Parameter)
    minimum_com_height: StorageType = default_storage_field(Parameter)
    minimum_feet_lateral_distance: StorageType = default_storage_field(Parameter)
    maximum_feet_relative_height: StorageType = default_storage_field(Parameter)
    maximum_joint_positions: StorageType = default_storage_field(Parameter)
    minimum_joint_positions: StorageType = default_storage_field(Parameter)
    maximum_joint_velocities: StorageType = default_storage_field(Parameter)
    minimum_joint_velocities: StorageType = default_storage_field(Parameter)
    references: CompositeType = default_composite_field(factory=References, time_varying=True)
    settings: dataclasses.InitVar[object] = dataclasses.field(default=None)
    kin_dyn_object: dataclasses.InitVar[object] = dataclasses.field(default=None)

    def __post_init__(self, settings, kin_dyn_object):
        nj = kin_dyn_object.NDoF
        self.system = ExtendedHumanoid(contact_point_descriptors=settings.contact_points, number_of_joints=nj)
        self.initial_state = ExtendedHumanoidState(contact_point_descriptors=settings.contact_points, number_of_joints=nj)
        self.final_state = hp_rp.HumanoidState(contact_point_descriptors=settings.contact_points, number_of_joints=nj)
        self.dt = settings.time_step
        self.gravity = np.asarray(getattr(kin_dyn_object, "g", settings.gravity), float)
        self.parametric_link_length_multipliers = 0.0
        self.parametric_link_densities = 0.0
        self.mass = kin_dyn_object.get_total_mass()
        for name in ("planar_dcc_height_multiplier", "dcc_gain", "dcc_epsilon", "static_friction", "maximum_velocity_control",
                     "maximum_force_derivative", "maximum_angular_momentum", "minimum_com_height", "minimum_feet_lateral_distance",
                     "maximum_feet_relative_height", "maximum_joint_positions", "minimum_joint_positions",
                     "maximum_joint_velocities", "minimum_joint_velocities"):
            setattr(self, name, getattr(settings, name))
        self.references = References(number_of_joints=nj, number_of_points_left=len(settings.contact_points.left),
                                     number_of_points_right=len(settings.contact_points.right))
