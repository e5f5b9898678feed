"""The `Variables` tree of the kinodynamic planner: 189 decision variables and 79 parameters per knot, 6 global variables, 326
global parameters.

The contract — node names, leaf names and order, storage kinds, which composites force their kind on their leaves — is the
reference's (turnkey_planners/humanoid_kinodynamic/variables.py:13-374) and IS the x / p layout of the engine; it is
pinned by tests/golden/kinodyn_structure.json, which the reference's own classes produced.  The declaration is table-driven
(`hippopt_amd/base/schema.py`): one `declare(...)` per node, defaults as data.
"""
import numpy as np

from ... import robot_planning as hp_rp
from ...base import Parameter, Variable
from ...base.schema import argument, child, declare, leaf

_M = __name__


def _column(count, value=0.0):
    return lambda: np.full((count, 1), value)


def _identity_quaternion():
    return np.array([[0.0], [0.0], [0.0], [1.0]])


# ---- references (parameters, one set per knot) -----------------------------------------------------------------------------------
def _contact_references_setup(self, number_of_points):
    self.desired_force_ratio = 1.0 / number_of_points if number_of_points else 0.0


ContactReferences = declare("ContactReferences", {"desired_force_ratio": leaf(Parameter), "number_of_points": argument()},
                            setup=_contact_references_setup, module=_M)


def _foot_references_setup(self, number_of_points):
    count = number_of_points or 0
    self.points = [ContactReferences(number_of_points=count) for _ in range(count)]
    self.yaw = 0.0


FootReferences = declare("FootReferences", {"points": child(list, time_varying=False), "yaw": leaf(Parameter), "number_of_points": argument(0)},
                         setup=_foot_references_setup, module=_M)


def _feet_references_setup(self, number_of_points_left, number_of_points_right):
    self.left, self.right = FootReferences(number_of_points=number_of_points_left), FootReferences(number_of_points=number_of_points_right)
    self.desired_swing_height = 0.02


FeetReferences = declare(
    "FeetReferences",
    {"left": child(lambda: FootReferences(), time_varying=False), "right": child(lambda: FootReferences(), time_varying=False),
     "desired_swing_height": leaf(Parameter), "number_of_points_left": argument(0), "number_of_points_right": argument(0)},
    setup=_feet_references_setup, module=_M)


def _references_setup(self, number_of_joints, number_of_points_left, number_of_points_right):
    self.feet = FeetReferences(number_of_points_left=number_of_points_left, number_of_points_right=number_of_points_right)
    for name, make in (("contacts_centroid_cost_weights", _column(3)), ("contacts_centroid", _column(3)), ("com_linear_velocity", _column(3)),
                       ("desired_frame_quaternion_xyzw", _identity_quaternion), ("base_quaternion_xyzw", _identity_quaternion),
                       ("base_quaternion_xyzw_velocity", _column(4)), ("joint_regularization", _column(number_of_joints))):
        setattr(self, name, make())


References = declare(
    "References",
    {"feet": child(lambda: FeetReferences(), time_varying=False),
     **{name: leaf(Parameter) for name in ("contacts_centroid_cost_weights", "contacts_centroid", "com_linear_velocity",
                                           "desired_frame_quaternion_xyzw", "base_quaternion_xyzw", "base_quaternion_xyzw_velocity",
                                           "joint_regularization")},
     "number_of_joints": argument(0), "number_of_points_left": argument(0), "number_of_points_right": argument(0)},
    setup=_references_setup, module=_M)


# ---- the system (variables, one set per knot) -------------------------------------------------------------------------------------
def _extended_point_setup(self, input_descriptor):
    hp_rp.ContactPointState.__post_init__(self, input_descriptor=input_descriptor)   # (also fills the derivative leaves: defaults of every base)
    self.u_v = np.zeros(3)


def _point_as_state(self):
    state = hp_rp.ContactPointState()
    state.p, state.f, state.descriptor = self.p, self.f, self.descriptor
    return state


ExtendedContactPoint = declare("ExtendedContactPoint", {"u_v": leaf(Variable)}, bases=(hp_rp.ContactPointState, hp_rp.ContactPointStateDerivative),
                               setup=_extended_point_setup, methods={"to_contact_point_state": _point_as_state}, module=_M)


def _feet_as_states(self):
    feet = hp_rp.FeetContactPoints()
    for side in ("left", "right"):
        setattr(feet, side, hp_rp.FootContactState.from_list([point.to_contact_point_state() for point in getattr(self, side)]))
    return feet


FeetContactPointsExtended = declare("FeetContactPointsExtended", {"left": child(list), "right": child(list)},
                                    methods={"to_feet_contact_points": _feet_as_states}, module=_M)


def _extended_humanoid_setup(self, contact_point_descriptors, number_of_joints):
    if contact_point_descriptors is not None:
        for side in ("left", "right"):
            setattr(self.contact_points, side, [ExtendedContactPoint(input_descriptor=d) for d in getattr(contact_point_descriptors, side)])
    self.com, self.centroidal_momentum = np.zeros(3), np.zeros(6)
    self.kinematics = hp_rp.FloatingBaseSystem(number_of_joints=number_of_joints)


def _humanoid_as_state(self):
    state = hp_rp.HumanoidState()
    state.kinematics, state.contact_points, state.com = (self.kinematics.to_floating_base_system_state(),
                                                          self.contact_points.to_feet_contact_points(), self.com)
    return state


ExtendedHumanoid = declare(
    "ExtendedHumanoid",
    {"contact_points": child(lambda: FeetContactPointsExtended()), "kinematics": child(hp_rp.FloatingBaseSystem, kind=Variable),
     "com": leaf(Variable), "centroidal_momentum": leaf(Variable), "contact_point_descriptors": argument(), "number_of_joints": argument()},
    setup=_extended_humanoid_setup, methods={"to_humanoid_state": _humanoid_as_state}, module=_M)


def _extended_state_setup(self, contact_point_descriptors, number_of_joints):
    hp_rp.HumanoidState.__post_init__(self, contact_point_descriptors=contact_point_descriptors, number_of_joints=number_of_joints)
    self.centroidal_momentum = np.zeros(6)


ExtendedHumanoidState = declare("ExtendedHumanoidState", {"centroidal_momentum": leaf(Variable)}, bases=(hp_rp.HumanoidState,),
                                setup=_extended_state_setup, module=_M)

# ---- the root ---------------------------------------------------------------------------------------------------------------------
# scalar / vector parameters copied from the settings object of the same name (variables.py:330-374)
_FROM_SETTINGS = ("planar_dcc_height_multiplier", "dcc_gain", "dcc_epsilon", "static_friction", "maximum_velocity_control",
                  "maximum_force_derivative", "maximum_angular_momentum", "minimum_com_height", "minimum_feet_lateral_distance",
                  "maximum_feet_relative_height", "maximum_joint_positions", "minimum_joint_positions", "maximum_joint_velocities",
                  "minimum_joint_velocities")


def _variables_setup(self, settings, kin_dyn_object):
    joints, feet = kin_dyn_object.NDoF, settings.contact_points
    self.system = ExtendedHumanoid(contact_point_descriptors=feet, number_of_joints=joints)
    self.initial_state = ExtendedHumanoidState(contact_point_descriptors=feet, number_of_joints=joints)
    self.final_state = hp_rp.HumanoidState(contact_point_descriptors=feet, number_of_joints=joints)
    self.mass = kin_dyn_object.get_total_mass()
    self.parametric_link_length_multipliers = self.parametric_link_densities = 0.0
    self.dt = settings.time_step
    self.gravity = np.asarray(getattr(kin_dyn_object, "g", settings.gravity), float)
    for name in _FROM_SETTINGS:
        setattr(self, name, getattr(settings, name))
    self.references = References(number_of_joints=joints, number_of_points_left=len(feet.left), number_of_points_right=len(feet.right))


Variables = declare(
    "Variables",
    {"system": child(lambda: ExtendedHumanoid(), kind=Variable),
     "mass": leaf(Parameter), "parametric_link_length_multipliers": leaf(Parameter), "parametric_link_densities": leaf(Parameter),
     "initial_state": child(lambda: ExtendedHumanoidState(), time_varying=False, kind=Parameter),
     "final_state": child(hp_rp.HumanoidState, time_varying=False, kind=Parameter),
     "dt": leaf(Parameter), "gravity": leaf(Parameter),
     **{name: leaf(Parameter) for name in _FROM_SETTINGS},
     "references": child(lambda: References(), time_varying=True),
     "settings": argument(), "kin_dyn_object": argument()},
    setup=_variables_setup, module=_M)
