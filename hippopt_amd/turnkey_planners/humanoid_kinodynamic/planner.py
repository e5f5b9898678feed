"""Planner — same surface as turnkey_planners/humanoid_kinodynamic/planner.py (constructor from Settings,
set_initial_guess / get_initial_guess / set_references / set_initial_state / set_final_state / solve /
get_variables_structure), with the engine-backed solver plugin in place of OptiSolver + CasADi."""
import copy

from ...base import MultipleShootingSolver, OptimalControlProblem, Output, extend_structure_to_horizon  # noqa: F401
from ...base.opti_callback import AcceptablePrimalInfeasibility, BestCost
from ...hipnlp_solver import HipNlpSolver
from .settings import Settings
from .variables import Variables


class Planner:
    def __init__(self, settings: Settings, model, device: int = 0, inner_solver: str = "auto", error_on_fail: bool = True) -> None:
        if not settings.is_valid():
            raise ValueError("Settings are not valid")
        self.settings = copy.deepcopy(settings)
        self.kin_dyn_object = model                       # the adam KinDynComputations counterpart (planner.py:43-50)
        self.numeric_mass = model.get_total_mass()
        variables = Variables(settings=self.settings, kin_dyn_object=model)
        opti_callback = None
        if self.settings.use_opti_callback:   # planner.py:56-63
            opti_callback = BestCost() & AcceptablePrimalInfeasibility(self.settings.acceptable_constraint_violation)
        self.optimization_solver = HipNlpSolver(self.settings, model, device=device, inner_solver=inner_solver,
                                                options_solver=self.settings.solver_options, callback_criterion=opti_callback,
                                                callback_save_costs=self.settings.opti_callback_save_costs,
                                                callback_save_constraint_multipliers=self.settings.opti_callback_save_constraint_multipliers,
                                                error_on_fail=error_on_fail)
        # the wiring of the reference's planner (planner.py:72-80), names kept: only the optimization solver differs
        self.solver = MultipleShootingSolver(optimization_solver=self.optimization_solver)
        self.ocp = OptimalControlProblem.create(input_structure=variables, optimal_control_solver=self.solver,
                                                horizon=self.settings.horizon_length)
        self.variables = self.optimization_solver.get_optimization_structure()

    # ---- mass regularisation (planner.py:932-1034): forces and momenta are divided by the total mass ---------
    def _scale_mass(self, var: Variables, factor: float) -> Variables:
        out = var
        if out.initial_state is not None:
            cm = out.initial_state.centroidal_momentum
            if cm is not None and hasattr(cm, "shape") and len(cm.shape) > 0 and cm.shape[0] == 6:
                out.initial_state.centroidal_momentum = cm * factor
            for point in out.initial_state.contact_points.left + out.initial_state.contact_points.right:
                point.f = point.f * factor
        if out.final_state is not None:
            for point in out.final_state.contact_points.left + out.final_state.contact_points.right:
                point.f = point.f * factor
        if out.system is None:
            return out
        for system in (out.system if isinstance(out.system, list) else [out.system]):
            if system.centroidal_momentum is not None:
                system.centroidal_momentum = system.centroidal_momentum * factor
            for point in system.contact_points.left + system.contact_points.right:
                if point.f is not None:
                    point.f = point.f * factor
        return out

    def _apply_mass_regularization(self, var):
        if self.numeric_mass == 0:
            raise ValueError("The mass of the robot is zero. This is not supported.")
        return self._scale_mass(var, 1.0 / self.numeric_mass)

    def _undo_mass_regularization(self, var):
        return self._scale_mass(var, self.numeric_mass)

    def set_initial_guess(self, initial_guess: Variables) -> None:
        self.optimization_solver.set_initial_guess(self._apply_mass_regularization(copy.deepcopy(initial_guess)))

    def get_initial_guess(self) -> Variables:
        return self._undo_mass_regularization(self.optimization_solver.get_initial_guess())

    def set_references(self, references) -> None:
        guess = self.optimization_solver.get_initial_guess()  # avoid the undo of the mass regularization (planner.py:1044-1057)
        assert isinstance(guess.references, list)
        assert not isinstance(references, list) or len(references) == len(guess.references)
        for i in range(len(guess.references)):
            guess.references[i] = references[i] if isinstance(references, list) else references
        self.optimization_solver.set_initial_guess(guess)

    def set_initial_state(self, initial_state) -> None:
        guess = self.get_initial_guess()
        guess.initial_state = initial_state
        self.set_initial_guess(guess)

    def set_final_state(self, final_state) -> None:
        guess = self.get_initial_guess()
        guess.final_state = final_state
        self.set_initial_guess(guess)

    def solve(self) -> Output:
        output = self.ocp.problem.solve()
        output.values = self._undo_mass_regularization(output.values)
        return output

    def get_variables_structure(self) -> Variables:
        return copy.deepcopy(self.variables)
