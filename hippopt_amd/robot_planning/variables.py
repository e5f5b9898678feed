"""Contact-point / floating-base / humanoid nodes of the variable tree: the per-knot memory layout of the kinodynamic NLP.

The CONTRACT is the reference's (robot_planning/variables/contacts.py:20-166, floating_base.py:16-185, humanoid.py:21-56): node
names, leaf names, storage kinds, constructor arguments and — decisive for the flat order — the field order of nodes with two
bases (derivative leaves first).  The DECLARATION is this build's: one `declare(...)` table per node (`hippopt_amd/base/schema.py`)
instead of a hand-written dataclass each; `tests/golden/kinodyn_structure.json`, produced by the reference's own classes, pins
that both give the same flat names, order, sizes and variable / parameter split (tests/test_structure.py).
"""
import copy
import dataclasses

import numpy as np

from ..base import OverridableVariable, Parameter
from ..base.schema import argument, child, declare, leaf, plain

_M = __name__


def _zeros(count):
    return lambda: np.zeros(count)


def _identity_quaternion():
    return np.array([0.0, 0.0, 0.0, 1.0])


# ---- contact points -------------------------------------------------------------------------------------------------------------
def _descriptor_setup(self, input_foot_frame, input_position_in_foot_frame):
    if input_foot_frame is not None:
        self.foot_frame = input_foot_frame
    if input_position_in_foot_frame is not None:
        self.position_in_foot_frame = input_position_in_foot_frame


def _rectangular_foot(foot_frame, x_length, y_length, top_left_point_position):
    """the four corners of a rectangular sole, counter-clockwise from the top-left one (contacts.py:38-65)"""
    corner = np.asarray(top_left_point_position, float)
    steps = np.array([[0.0, 0.0], [-1.0, 0.0], [-1.0, -1.0], [0.0, -1.0]]) * [x_length, y_length]
    return [ContactPointDescriptor(input_foot_frame=foot_frame, input_position_in_foot_frame=corner + np.append(step, 0.0)) for step in steps]


ContactPointDescriptor = declare(
    "ContactPointDescriptor",
    {"position_in_foot_frame": leaf(Parameter), "foot_frame": plain(), "input_foot_frame": argument(), "input_position_in_foot_frame": argument()},
    setup=_descriptor_setup, methods={"rectangular_foot": staticmethod(_rectangular_foot)}, module=_M)


def _point_setup(self, input_descriptor):
    if input_descriptor is not None:
        self.descriptor = copy.deepcopy(input_descriptor)


ContactPointState = declare(
    "ContactPointState",
    {"p": leaf(OverridableVariable, _zeros(3)), "f": leaf(OverridableVariable, _zeros(3)),
     "descriptor": child(lambda: ContactPointDescriptor(), time_varying=False), "input_descriptor": argument()},
    setup=_point_setup, module=_M)

ContactPointStateDerivative = declare(
    "ContactPointStateDerivative", {"v": leaf(OverridableVariable, _zeros(3)), "f_dot": leaf(OverridableVariable, _zeros(3))}, module=_M)


@dataclasses.dataclass
class FootContactState(list, ContactPointState.__mro__[1]):
    """the contact points of one foot (a list that is also a node: contacts.py:97-127)"""

    def set_from_parent_frame_transform(self, transform):
        origin, turn = transform.translation(), transform.rotation()
        for point in self:
            point.p = origin + turn.act(point.descriptor.position_in_foot_frame)

    @staticmethod
    def from_list(input_list):
        foot = FootContactState()
        foot.extend(input_list)
        return foot

    @staticmethod
    def from_parent_frame_transform(descriptor, transform):
        foot = FootContactState.from_list([ContactPointState(input_descriptor=d) for d in descriptor])
        foot.set_from_parent_frame_transform(transform)
        return foot


@dataclasses.dataclass
class _LeftRight:
    left: list = dataclasses.field(default_factory=list)
    right: list = dataclasses.field(default_factory=list)


class FeetContactPointDescriptors(_LeftRight):
    """descriptors of the contact points of the two feet"""


class FeetContactPhasesDescriptor(_LeftRight):
    """contact phase lists of the two feet (contacts.py:163-166)"""


@dataclasses.dataclass
class FootContactPhaseDescriptor:
    """one stance of a foot (contacts.py:142-160): where, with which force, from when to when; `mid_swing_transform`: the pose
    half way through the swing that FOLLOWS this stance (None: half way to the next stance, with its orientation)"""
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
            self.force = np.array([0.0, 0.0, 100.0])


FeetContactPoints = declare("FeetContactPoints", {"left": child(FootContactState), "right": child(FootContactState)}, module=_M)

# ---- floating base, joints ------------------------------------------------------------------------------------------------------
FreeFloatingObjectState = declare(
    "FreeFloatingObjectState", {"position": leaf(OverridableVariable, _zeros(3)), "quaternion_xyzw": leaf(OverridableVariable, _identity_quaternion)},
    module=_M)
FreeFloatingObjectStateDerivative = declare(
    "FreeFloatingObjectStateDerivative",
    {"linear_velocity": leaf(OverridableVariable, _zeros(3)), "quaternion_velocity_xyzw": leaf(OverridableVariable, _zeros(4))}, module=_M)
FreeFloatingObject = declare("FreeFloatingObject", {}, bases=(FreeFloatingObjectState, FreeFloatingObjectStateDerivative), module=_M)


def _tree_state_setup(self, number_of_joints_state):
    if number_of_joints_state is not None and self.positions is None:
        self.positions = np.zeros(number_of_joints_state)


def _tree_derivative_setup(self, number_of_joints_derivative):
    if number_of_joints_derivative is not None:
        self.velocities = np.zeros(number_of_joints_derivative)


def _tree_setup(self, number_of_joints_derivative=None, number_of_joints_state=None):
    counts = [c for c in (number_of_joints_state, number_of_joints_derivative) if c is not None]
    if counts:   # one count serves both halves
        _tree_state_setup(self, counts[0])
        _tree_derivative_setup(self, counts[-1])


KinematicTreeState = declare("KinematicTreeState", {"positions": leaf(OverridableVariable), "number_of_joints_state": argument(0)},
                             setup=_tree_state_setup, module=_M)
KinematicTreeStateDerivative = declare("KinematicTreeStateDerivative",
                                       {"velocities": leaf(OverridableVariable), "number_of_joints_derivative": argument()},
                                       setup=_tree_derivative_setup, module=_M)
KinematicTree = declare("KinematicTree", {}, bases=(KinematicTreeState, KinematicTreeStateDerivative), setup=_tree_setup, module=_M)


def _system_state_setup(self, number_of_joints_state):
    if number_of_joints_state is not None:
        self.joints = KinematicTreeState(number_of_joints_state=number_of_joints_state)


def _system_setup(self, number_of_joints):
    if number_of_joints is not None:
        self.joints = KinematicTree(number_of_joints_state=number_of_joints)


def _system_as_state(self):
    """positions only (floating_base.py:179-185): the state half of a system that also carries velocities"""
    state = FloatingBaseSystemState()
    state.base.position, state.base.quaternion_xyzw, state.joints.positions = self.base.position, self.base.quaternion_xyzw, self.joints.positions
    return state


FloatingBaseSystemState = declare(
    "FloatingBaseSystemState",
    {"base": child(lambda: FreeFloatingObjectState()), "joints": child(lambda: KinematicTreeState()), "number_of_joints_state": argument()},
    setup=_system_state_setup, module=_M)
FloatingBaseSystem = declare(
    "FloatingBaseSystem", {"base": child(lambda: FreeFloatingObject()), "joints": child(lambda: KinematicTree()), "number_of_joints": argument()},
    setup=_system_setup, methods={"to_floating_base_system_state": _system_as_state}, module=_M)


# ---- humanoid -------------------------------------------------------------------------------------------------------------------
def _humanoid_setup(self, contact_point_descriptors, number_of_joints):
    if contact_point_descriptors is not None:
        for side in ("left", "right"):
            setattr(self.contact_points, side, [ContactPointState(input_descriptor=d) for d in getattr(contact_point_descriptors, side)])
    if number_of_joints is not None:
        self.kinematics = FloatingBaseSystemState(number_of_joints_state=number_of_joints)


HumanoidState = declare(
    "HumanoidState",
    {"contact_points": child(lambda: FeetContactPoints(), time_varying=False), "kinematics": child(lambda: FloatingBaseSystemState(), time_varying=False),
     "com": leaf(OverridableVariable, _zeros(3)), "contact_point_descriptors": argument(), "number_of_joints": argument()},
    setup=_humanoid_setup, module=_M)
