"""Contact-point / floating-base / humanoid dataclasses: the per-knot memory layout of the kinodynamic NLP.
Mirror of robot_planning/variables/contacts.py:20-166, floating_base.py:16-185, humanoid.py:21-56 (field names, storage
types and — decisive for the flat order — the multiple-inheritance field order: derivative fields first)."""
import copy
import dataclasses

import numpy as np

from ..base import (CompositeType, OptimizationObject, OverridableVariable, Parameter, StorageType, default_composite_field,
                    default_storage_field)


@dataclasses.dataclass
class ContactPointDescriptor(OptimizationObject):
    position_in_foot_frame: StorageType = default_storage_field(Parameter)
    foot_frame: str = dataclasses.field(default=None)
    input_foot_frame: dataclasses.InitVar[str] = dataclasses.field(default=None)
    input_position_in_foot_frame: dataclasses.InitVar[np.ndarray] = dataclasses.field(default=None)

    def __post_init__(self, input_foot_frame, input_position_in_foot_frame):
        if input_foot_frame is not None:
            self.foot_frame = input_foot_frame
        if input_position_in_foot_frame is not None:
            self.position_in_foot_frame = input_position_in_foot_frame

    @staticmethod
    def rectangular_foot(foot_frame, x_length, y_length, top_left_point_position):
        tl = np.asarray(top_left_point_position, float)
        return [ContactPointDescriptor(input_foot_frame=foot_frame, input_position_in_foot_frame=tl + d)
                for d in ([0.0, 0.0, 0.0], [-x_length, 0.0, 0.0], [-x_length, -y_length, 0.0], [0.0, -y_length, 0.0])]


@dataclasses.dataclass
class ContactPointState(OptimizationObject):
    p: StorageType = default_storage_field(OverridableVariable)
    f: StorageType = default_storage_field(OverridableVariable)
    descriptor: CompositeType = default_composite_field(factory=ContactPointDescriptor, time_varying=False)
    input_descriptor: dataclasses.InitVar[ContactPointDescriptor] = dataclasses.field(default=None)

    def __post_init__(self, input_descriptor):
        self.p = np.zeros(3) if self.p is None else self.p
        self.f = np.zeros(3) if self.f is None else self.f
        if input_descriptor is not None:
            self.descriptor = copy.deepcopy(input_descriptor)


@dataclasses.dataclass
class ContactPointStateDerivative(OptimizationObject):
    v: StorageType = default_storage_field(OverridableVariable)
    f_dot: StorageType = default_storage_field(OverridableVariable)

    def __post_init__(self):
        self.v = np.zeros(3) if self.v is None else self.v
        self.f_dot = np.zeros(3) if self.f_dot is None else self.f_dot


@dataclasses.dataclass
class FootContactState(list, OptimizationObject):
    def set_from_parent_frame_transform(self, transform):   # contacts.py:103-107
        for contact_point in self:
            contact_point.p = transform.translation() + transform.rotation().act(contact_point.descriptor.position_in_foot_frame)

    @staticmethod
    def from_list(input_list):
        out = FootContactState()
        out.extend(input_list)
        return out

    @staticmethod
    def from_parent_frame_transform(descriptor, transform):   # contacts.py:116-127
        out = FootContactState()
        for contact_point_descriptor in descriptor:
            out.append(ContactPointState(input_descriptor=contact_point_descriptor))
        out.set_from_parent_frame_transform(transform)
        return out


@dataclasses.dataclass
class FeetContactPointDescriptors:
    left: list = dataclasses.field(default_factory=list)
    right: list = dataclasses.field(default_factory=list)


@dataclasses.dataclass
class FootContactPhaseDescriptor:   # contacts.py:142-160
    transform: object = None
    mid_swing_transform: object = None
    force: np.ndarray = None
    activation_time: float = None
    deactivation_time: float = None

    def __post_init__(self):
        from .transforms import SE3, SO3
        if self.transform is None:
            self.transform = SE3.from_translation_and_rotation(np.zeros(3), SO3.Identity())
        if self.force is None:
            self.force = np.zeros(3)
            self.force[2] = 100


@dataclasses.dataclass
class FeetContactPhasesDescriptor:   # contacts.py:163-166
    left: list = dataclasses.field(default_factory=list)
    right: list = dataclasses.field(default_factory=list)


@dataclasses.dataclass
class FeetContactPoints(OptimizationObject):
    left: list = default_composite_field(factory=FootContactState)
    right: list = default_composite_field(factory=FootContactState)


@dataclasses.dataclass
class FreeFloatingObjectState(OptimizationObject):
    position: StorageType = default_storage_field(OverridableVariable)
    quaternion_xyzw: StorageType = default_storage_field(OverridableVariable)

    def __post_init__(self):
        self.position = np.zeros(3) if self.position is None else self.position
        if self.quaternion_xyzw is None:
            self.quaternion_xyzw = np.array([0.0, 0.0, 0.0, 1.0])


@dataclasses.dataclass
class FreeFloatingObjectStateDerivative(OptimizationObject):
    linear_velocity: StorageType = default_storage_field(OverridableVariable)
    quaternion_velocity_xyzw: StorageType = default_storage_field(OverridableVariable)

    def __post_init__(self):
        self.linear_velocity = np.zeros(3) if self.linear_velocity is None else self.linear_velocity
        self.quaternion_velocity_xyzw = np.zeros(4) if self.quaternion_velocity_xyzw is None else self.quaternion_velocity_xyzw


@dataclasses.dataclass
class FreeFloatingObject(FreeFloatingObjectState, FreeFloatingObjectStateDerivative):
    def __post_init__(self):
        FreeFloatingObjectState.__post_init__(self)
        FreeFloatingObjectStateDerivative.__post_init__(self)


@dataclasses.dataclass
class KinematicTreeState(OptimizationObject):
    positions: StorageType = default_storage_field(OverridableVariable)
    number_of_joints_state: dataclasses.InitVar[int] = dataclasses.field(default=0)

    def __post_init__(self, number_of_joints_state):
        if number_of_joints_state is not None and self.positions is None:
            self.positions = np.zeros(number_of_joints_state)


@dataclasses.dataclass
class KinematicTreeStateDerivative(OptimizationObject):
    velocities: StorageType = default_storage_field(OverridableVariable)
    number_of_joints_derivative: dataclasses.InitVar[int] = dataclasses.field(default=None)

    def __post_init__(self, number_of_joints_derivative):
        if number_of_joints_derivative is not None:
            self.velocities = np.zeros(number_of_joints_derivative)


@dataclasses.dataclass
class KinematicTree(KinematicTreeState, KinematicTreeStateDerivative):
    def __post_init__(self, number_of_joints_derivative=None, number_of_joints_state=None):
        if number_of_joints_derivative is not None or number_of_joints_state is not None:
            ns = number_of_joints_derivative if number_of_joints_state is None else number_of_joints_state
            nd = ns if number_of_joints_derivative is None else number_of_joints_derivative
            KinematicTreeState.__post_init__(self, number_of_joints_state=ns)
            KinematicTreeStateDerivative.__post_init__(self, number_of_joints_derivative=nd)


@dataclasses.dataclass
class FloatingBaseSystemState(OptimizationObject):
    base: CompositeType = default_composite_field(factory=FreeFloatingObjectState)
    joints: CompositeType = default_composite_field(factory=KinematicTreeState)
    number_of_joints_state: dataclasses.InitVar[int] = dataclasses.field(default=None)

    def __post_init__(self, number_of_joints_state):
        if number_of_joints_state is not None:
            self.joints = KinematicTreeState(number_of_joints_state=number_of_joints_state)


@dataclasses.dataclass
class FloatingBaseSystem(OptimizationObject):
    base: CompositeType = default_composite_field(factory=FreeFloatingObject)
    joints: CompositeType = default_composite_field(factory=KinematicTree)
    number_of_joints: dataclasses.InitVar[int] = dataclasses.field(default=None)

    def __post_init__(self, number_of_joints):
        if number_of_joints is not None:
            self.joints = KinematicTree(number_of_joints_state=number_of_joints)

    def to_floating_base_system_state(self):
        out = FloatingBaseSystemState()
        out.base.position = self.base.position
        out.base.quaternion_xyzw = self.base.quaternion_xyzw
        out.joints.positions = self.joints.positions
        return out


@dataclasses.dataclass
class HumanoidState(OptimizationObject):
    contact_points: CompositeType = default_composite_field(factory=FeetContactPoints, time_varying=False)
    kinematics: CompositeType = default_composite_field(factory=FloatingBaseSystemState, time_varying=False)
    com: StorageType = default_storage_field(OverridableVariable)
    contact_point_descriptors: dataclasses.InitVar[FeetContactPointDescriptors] = dataclasses.field(default=None)
    number_of_joints: dataclasses.InitVar[int] = dataclasses.field(default=None)

    def __post_init__(self, contact_point_descriptors, number_of_joints):
        if contact_point_descriptors is not None:
            self.contact_points.left = [ContactPointState(input_descriptor=pt) for pt in contact_point_descriptors.left]
            self.contact_points.right = [ContactPointState(input_descriptor=pt) for pt in contact_point_descriptors.right]
        if number_of_joints is not None:
            self.kinematics = FloatingBaseSystemState(number_of_joints_state=number_of_joints)
        self.com = np.zeros(3) if self.com is None else self.com
