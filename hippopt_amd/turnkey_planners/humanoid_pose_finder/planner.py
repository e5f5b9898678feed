"""Mirror of turnkey_planners/humanoid_pose_finder/planner.py: Settings (:19-193), References (:196-226), Variables (:229-320)
and the Planner surface (:323-955: set_initial_guess / get_initial_guess / set_references / solve / get_variables_structure,
mass regularisation of the forces), with the engine-backed solver plugin in place of OptiSolver + CasADi.

The hand position expressions (planner.py:62-69, 596-660) take the reference's frame NAMES (`left_hand_frame_name`,
`right_hand_frame_name`), resolved through the model's named frames (a URDF's links, `RobotModel.resolve_frame`).
Not mirrored: the parametric-link model (adam.parametric), see include/hipnlp.h."""
import copy
import dataclasses

import numpy as np

from ... import robot_planning as hp_rp
from ...base import (CompositeType, OptimizationObject, Output, Parameter, StorageType, Variable, default_composite_field,
                     default_storage_field)
from ...hipnlp_solver import HipNlpSolver
from ...pose_settings import PoseSettings


@dataclasses.dataclass
class Settings(PoseSettings):
    joints_name_list: list = None
    root_link: str = "root_link"
    contact_points: hp_rp.FeetContactPointDescriptors = None
    desired_frame_quaternion_cost_frame_name: str = "chest"
    left_hand_frame_name: str = None
    right_hand_frame_name: str = None
    solver_options: dict = dataclasses.field(default_factory=dict)

    def __post_init__(self):
        PoseSettings.__post_init__(self)
        if self.contact_points is None:
            self.contact_points = hp_rp.FeetContactPointDescriptors()
            self.contact_points.left = hp_rp.ContactPointDescriptor.rectangular_foot("l_sole", 0.232, 0.1, [0.116, 0.05, 0.0])
            self.contact_points.right = hp_rp.ContactPointDescriptor.rectangular_foot("r_sole", 0.232, 0.1, [0.116, 0.05, 0.0])

    def is_valid(self) -> bool:
        from ... import _abi
        ok = (self.gravity is not None and len(self.gravity) == 6 and self.maximum_joint_positions is not None
              and self.minimum_joint_positions is not None and self.joint_regularization_cost_weights is not None)
        for mode, name, frame in ((self.left_hand_expression_type, self.left_hand_frame_name, self.left_hand_frame),
                                  (self.right_hand_expression_type, self.right_hand_frame_name, self.right_hand_frame)):
            if mode is None or (mode != _abi.EXPR_SKIP and name is None and frame is None):   # planner.py:162-186
                ok = False
        return ok


@dataclasses.dataclass
class References(OptimizationObject):
    state: CompositeType = default_composite_field(cls=Parameter, factory=hp_rp.HumanoidState)
    frame_quaternion_xyzw: StorageType = default_storage_field(Parameter)
    left_hand_position: StorageType = default_storage_field(Parameter)
    right_hand_position: StorageType = default_storage_field(Parameter)
    contact_point_descriptors: dataclasses.InitVar[hp_rp.FeetContactPointDescriptors] = dataclasses.field(default=None)
    number_of_joints: dataclasses.InitVar[int] = dataclasses.field(default=None)

    def __post_init__(self, contact_point_descriptors, number_of_joints):
        self.state = hp_rp.HumanoidState(contact_point_descriptors=contact_point_descriptors, number_of_joints=number_of_joints)
        self.frame_quaternion_xyzw = np.array([0.0, 0.0, 0.0, 1.0])
        self.left_hand_position = np.zeros(3)
        self.right_hand_position = np.zeros(3)


@dataclasses.dataclass
class Variables(OptimizationObject):
    state: CompositeType = default_composite_field(cls=Variable, factory=hp_rp.HumanoidState)
    mass: StorageType = default_storage_field(Parameter)
    parametric_link_length_multipliers: StorageType = default_storage_field(Parameter)
    parametric_link_densities: StorageType = default_storage_field(Parameter)
    gravity: StorageType = default_storage_field(Parameter)
    references: CompositeType = default_composite_field(cls=Parameter, factory=References)
    relaxed_complementarity_epsilon: StorageType = default_storage_field(Parameter)
    static_friction: StorageType = default_storage_field(Parameter)
    maximum_joint_positions: StorageType = default_storage_field(Parameter)
    minimum_joint_positions: StorageType = default_storage_field(Parameter)
    left_hand_position_in_frame: StorageType = default_storage_field(Parameter)
    right_hand_position_in_frame: StorageType = default_storage_field(Parameter)
    settings: dataclasses.InitVar[object] = dataclasses.field(default=None)
    kin_dyn_object: dataclasses.InitVar[object] = dataclasses.field(default=None)

    def __post_init__(self, settings, kin_dyn_object):
        nj = kin_dyn_object.NDoF
        self.state = hp_rp.HumanoidState(contact_point_descriptors=settings.contact_points, number_of_joints=nj)
        self.references = References(contact_point_descriptors=settings.contact_points, number_of_joints=nj)
        self.parametric_link_length_multipliers = 0.0   # non-parametric model (planner.py:266-270, :286-288)
        self.parametric_link_densities = 0.0
        self.mass = kin_dyn_object.get_total_mass()
        self.gravity = np.asarray(settings.gravity, float)
        self.static_friction = settings.static_friction
        self.relaxed_complementarity_epsilon = settings.relaxed_complementarity_epsilon
        self.maximum_joint_positions = settings.maximum_joint_positions
        self.minimum_joint_positions = settings.minimum_joint_positions
        self.left_hand_position_in_frame = np.asarray(settings.lef_hand_position_in_frame, float)      # planner.py:298-299
        self.right_hand_position_in_frame = np.asarray(settings.right_hand_position_in_frame, float)


class Planner:
    def __init__(self, settings: Settings, model, device: int = 0, inner_solver: str = "auto", error_on_fail: bool = True) -> None:
        if not settings.is_valid():
            raise ValueError("Settings are not valid")
        self.settings = copy.deepcopy(settings)
        for side in ("left", "right"):   # frame names -> (link, link_T_frame) of the model
            name = getattr(self.settings, side + "_hand_frame_name")
            if name is not None and getattr(self.settings, side + "_hand_frame") is None:
                setattr(self.settings, side + "_hand_frame", model.resolve_frame(name))
        self.kin_dyn_object = model
        self.numeric_mass = model.get_total_mass()
        self.variables = Variables(settings=self.settings, kin_dyn_object=model)
        self.optimization_solver = HipNlpSolver(self.settings, model, device=device, inner_solver=inner_solver,
                                                options_solver=self.settings.solver_options, problem="pose", error_on_fail=error_on_fail)
        self.optimization_solver.generate_optimization_objects(self.variables)

    # ---- mass regularisation (planner.py:788-850): contact forces of the state and of the references / mass ----------
    def _scale_forces(self, var: Variables, factor: float) -> Variables:
        if self.numeric_mass == 0:
            raise ValueError("The mass of the robot is zero. This is not supported.")
        out = var
        for holder in (out.state, out.references.state if out.references is not None else None):
            if holder is None:
                continue
            for point in holder.contact_points.left + holder.contact_points.right:
                point.f = np.asarray(point.f, float) * factor
        return out

    def _apply_mass_regularization(self, var):
        return self._scale_forces(var, 1.0 / self.numeric_mass)

    def _undo_mass_regularization(self, var):
        return self._scale_forces(var, self.numeric_mass)

    def set_initial_guess(self, initial_guess: Variables) -> None:
        self.optimization_solver.set_initial_guess(self._apply_mass_regularization(copy.deepcopy(initial_guess)))

    def get_initial_guess(self) -> Variables:
        return self._undo_mass_regularization(self.optimization_solver.get_initial_guess())

    def set_references(self, references: References) -> None:
        guess = self.optimization_solver.get_initial_guess()  # avoid the undo of the mass regularization (planner.py:859-864)
        guess.references = copy.deepcopy(references)
        self.set_initial_guess(guess)

    def solve(self) -> Output:
        s = self.optimization_solver
        s.solve()
        return Output(values=self._undo_mass_regularization(s.get_values()), cost_value=s.get_cost_value(),
                      cost_values=s.get_cost_values(), constraint_multipliers=s.get_constraint_multipliers())

    def get_variables_structure(self) -> Variables:
        return copy.deepcopy(self.variables)
