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
    def __init__(self, settings: Settings, model, device: int = 0, inner_solver: str = "auto", error_on_fail: bool = True, devices=None) -> None:
        """devices: HIP ordinals — the horizon is cut into one knot range per entry, every callback of the one NLP driver is evaluated by
        all of them (HipNlpSolver(devices=...)); None: one handle on `device`"""
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
                                                error_on_fail=error_on_fail, devices=devices)
        # the wiring of the reference's planner (planner.py:72-80), names kept: only the optimization solver differs
        self.solver = MultipleShootingSolver(optimization_solver=self.optimization_solver)
        self.ocp = OptimalControlProblem.create(input_structure=variables, optimal_control_solver=self.solver,
                                                horizon=self.settings.horizon_length)
        self.variables = self.optimization_solver.get_optimization_structure()

    # ---- mass regularisation (planner.py:932-1034): the solver works with forces and momenta divided by the total mass --------------
    @staticmethod
    def _mass_scaled(var: Variables):
        """(owner, attribute) of every quantity of a variables tree that carries a factor of the robot's mass"""
        ends = [state for state in (var.initial_state, var.final_state) if state is not None]
        knots = [] if var.system is None else (var.system if isinstance(var.system, list) else [var.system])
        for holder, optional in [(e, False) for e in ends] + [(k, True) for k in knots]:
            for point in holder.contact_points.left + holder.contact_points.right:
                if point.f is not None or not optional:      # (a state always carries its forces; a knot of a partial guess may not)
                    yield point, "f"
        momentum = None if var.initial_state is None else var.initial_state.centroidal_momentum
        if momentum is not None and getattr(momentum, "shape", ()) and momentum.shape[0] == 6:
            yield var.initial_state, "centroidal_momentum"
        for knot in knots:
            if knot.centroidal_momentum is not None:
                yield knot, "centroidal_momentum"

    def _rescaled(self, var: Variables, factor: float) -> Variables:
        for owner, name in self._mass_scaled(var):
            setattr(owner, name, getattr(owner, name) * factor)
        return var

    def _apply_mass_regularization(self, var):
        if self.numeric_mass == 0:
            raise ValueError("The mass of the robot is zero. This is not supported.")
        return self._rescaled(var, 1.0 / self.numeric_mass)

    def _undo_mass_regularization(self, var):
        return self._rescaled(var, self.numeric_mass)

    # ---- guesses: the plugin stores the REGULARISED tree; everything a user hands over or gets back is in physical units ----------------
    def set_initial_guess(self, initial_guess: Variables) -> None:
        self.optimization_solver.set_initial_guess(self._apply_mass_regularization(copy.deepcopy(initial_guess)))

    def get_initial_guess(self) -> Variables:
        return self._undo_mass_regularization(self.optimization_solver.get_initial_guess())

    def _replace_in_guess(self, field: str, value) -> None:
        guess = self.get_initial_guess()
        setattr(guess, field, value)
        self.set_initial_guess(guess)

    def set_initial_state(self, initial_state) -> None:
        self._replace_in_guess("initial_state", initial_state)

    def set_final_state(self, final_state) -> None:
        self._replace_in_guess("final_state", final_state)

    def set_references(self, references) -> None:
        """one reference for every knot, or a list with one per knot; references carry no mass factor, so the stored (regularised)
        tree is edited in place of a round trip through physical units (planner.py:1044-1057)"""
        stored = self.optimization_solver.get_initial_guess()
        slots = stored.references
        if not isinstance(slots, list):
            raise TypeError("the references of the expanded structure are one entry per knot")
        per_knot = isinstance(references, list)
        if per_knot and len(references) != len(slots):
            raise ValueError("%d references for %d knots" % (len(references), len(slots)))
        slots[:] = list(references) if per_knot else [references] * len(slots)
        self.optimization_solver.set_initial_guess(stored)

    def solve(self) -> Output:
        output = self.ocp.problem.solve()
        output.values = self._undo_mass_regularization(output.values)
        return output

    def get_variables_structure(self) -> Variables:
        return copy.deepcopy(self.variables)
