"""The planner-level facade with the reference's names: `OptimizationSolver` (the plugin interface, base/optimization_solver.py:25-96),
`MultipleShootingSolver(optimization_solver=...)` (base/multiple_shooting_solver.py:28-55) and
`OptimalControlProblem.create(input_structure=..., optimal_control_solver=..., horizon=N)` (base/optimal_control_problem.py:66-83),
wired as the reference's planner wires them (turnkey_planners/humanoid_kinodynamic/planner.py:65-80):

    solver = MultipleShootingSolver(optimization_solver=HipNlpSolver(settings, model, ...))
    ocp = OptimalControlProblem.create(input_structure=variables, optimal_control_solver=solver, horizon=settings.horizon_length)
    ...
    output = ocp.problem.solve()

What differs, and why: in the reference the planner then DESCRIBES the NLP to this layer as CasADi expressions (`add_dynamics`,
`add_expression_to_horizon`, `add_cost`, `add_constraint`, planner.py:124-176) and CasADi's VM evaluates them on one CPU thread.
Here that list is the typed row / cost directory built into the engine (`hippopt_amd/csrc/layout.h`, include/hipnlp.h) — an
engine-backed solver refuses symbolic expressions with a message that says so — while everything a caller does around it keeps
its name and meaning: structure expansion over the horizon, guesses, `initial` / `final`, `solve()` returning `Output`, per-named
costs and multipliers.  The small closed-form OCPs of the reference's own tests (affine dynamics, quadratic costs) are served by the
CPU plugin `hippopt_amd.base.affine_solver.AffineSolver`, which does accept expressions (`hippopt_amd/base/affine.py`).
"""
import abc
import copy

import numpy as np

from .horizon import extend_structure_to_horizon, flattened_names
from .problem import Output, ProblemNotSolvedException


class OptimizationSolver(abc.ABC):
    """The plugin interface a solver offers to the layers above (method set of base/optimization_solver.py:25-96)."""

    @abc.abstractmethod
    def generate_optimization_objects(self, input_structure, **kwargs): ...

    @abc.abstractmethod
    def get_optimization_objects(self): ...

    @abc.abstractmethod
    def get_optimization_structure(self): ...

    @abc.abstractmethod
    def register_problem(self, problem): ...

    @abc.abstractmethod
    def get_problem(self): ...

    @abc.abstractmethod
    def set_initial_guess(self, initial_guess): ...

    @abc.abstractmethod
    def get_initial_guess(self): ...

    @abc.abstractmethod
    def solve(self): ...

    @abc.abstractmethod
    def get_values(self): ...

    @abc.abstractmethod
    def get_cost_value(self): ...

    @abc.abstractmethod
    def add_cost(self, input_cost, name=None): ...

    @abc.abstractmethod
    def add_constraint(self, input_constraint, name=None): ...

    @abc.abstractmethod
    def cost_function(self): ...

    @abc.abstractmethod
    def get_cost_expressions(self): ...

    @abc.abstractmethod
    def get_constraint_expressions(self): ...

    @abc.abstractmethod
    def get_cost_values(self): ...

    @abc.abstractmethod
    def get_constraint_multipliers(self): ...


class TypedProblemError(NotImplementedError):
    """a symbolic expression was handed to a solver whose problem is the typed list built into the engine"""

    def __init__(self, what):
        super().__init__(
            what + ": this solver evaluates the kinodynamic / pose-finder NLP of the reference's planners as the typed row and cost "
            "directory built into the engine (hippopt_amd/csrc/layout.h; settings select the expression types and weights). "
            "Problems described by expressions need a solver that accepts them: hippopt_amd.base.affine_solver.AffineSolver for "
            "affine dynamics / quadratic costs on the CPU, or the reference's OptiSolver for general CasADi graphs.")


def _accepts_expressions(solver):
    return bool(getattr(solver, "accepts_expressions", False))


class MultipleShootingSolver:
    """Transcription layer between a problem and an `OptimizationSolver` (base/multiple_shooting_solver.py): expands the structure over
    the horizon, knows every flattened variable by name, hands guesses / solve / results through."""

    def __init__(self, optimization_solver=None, default_integrator=None):
        if optimization_solver is None:
            raise ValueError("MultipleShootingSolver needs an optimization_solver (the reference defaults to CasADi's OptiSolver, "
                             "which this build does not ship): HipNlpSolver(...) or AffineSolver()")
        for name in ("generate_optimization_objects", "set_initial_guess", "solve", "get_values"):
            if not callable(getattr(optimization_solver, name, None)):
                raise TypeError("optimization_solver does not offer the OptimizationSolver interface: no " + name + "()")
        self._optimization_solver = optimization_solver
        self._default_integrator = default_integrator
        self._symbolic_structure = None
        self._names = {}
        self._problem = None

    # ---- structure -----------------------------------------------------------------------------------------------------------
    def generate_optimization_objects(self, input_structure, **kwargs):
        objects = self._optimization_solver.generate_optimization_objects(input_structure=input_structure, **kwargs)
        self._names = flattened_names(extend_structure_to_horizon(input_structure, **kwargs), input_structure) \
            if not isinstance(input_structure, list) else {}
        if _accepts_expressions(self._optimization_solver):
            self._symbolic_structure = self._optimization_solver.symbolic_structure(input_structure, self._names)
        else:
            self._symbolic_structure = copy.deepcopy(input_structure)   # (names and shapes; no expression graph behind it)
        return objects

    def get_optimization_objects(self):
        return self._optimization_solver.get_optimization_objects()

    def get_optimization_structure(self):
        return self._optimization_solver.get_optimization_structure()

    def get_symbolic_structure(self):
        return self._symbolic_structure

    def get_optimization_solver(self):
        return self._optimization_solver

    def register_problem(self, problem):
        self._problem = problem
        self._optimization_solver.register_problem(problem)

    def get_problem(self):
        return self._problem

    # ---- the horizon --------------------------------------------------------------------------------------------------------
    def _knot_names(self, variable):
        name = getattr(variable, "flat_name", variable)
        if not isinstance(name, str) or name not in self._names:
            raise ValueError("Variable " + str(name) + " not found in the optimization variables.")
        return self._names[name][1]

    def _element(self, variable, which):
        names = self._knot_names(variable)
        if _accepts_expressions(self._optimization_solver):
            return self._optimization_solver.symbol(names[which])
        return self._optimization_solver.get_optimization_objects().to_dict()[names[which]]

    def initial(self, variable):
        """first element of a flattened variable (multiple_shooting_solver.py:826-849): an expression for a solver that accepts
        them, the value held for it otherwise"""
        return self._element(variable, 0)

    def final(self, variable):
        return self._element(variable, -1)

    def add_dynamics(self, dynamics, x0=None, t0=0.0, mode=None, name=None, x0_name=None, **kwargs):
        if not _accepts_expressions(self._optimization_solver):
            raise TypedProblemError("add_dynamics")
        from .affine import add_dynamics_to_horizon
        add_dynamics_to_horizon(self, dynamics, x0=x0, t0=t0, mode=mode, name=name, x0_name=x0_name,
                                default_integrator=self._default_integrator, **kwargs)

    def add_expression_to_horizon(self, expression, mode=None, apply_to_first_elements=False, name=None, **kwargs):
        if not _accepts_expressions(self._optimization_solver):
            raise TypedProblemError("add_expression_to_horizon")
        from .affine import add_expression_to_horizon
        add_expression_to_horizon(self, expression, mode=mode, apply_to_first_elements=apply_to_first_elements, name=name, **kwargs)


def _hand_through(method):
    """a method of the transcription layer that is the plugin's method of the same name (same arguments, same result)"""
    def forward(self, *args, **kwargs):
        return getattr(self._optimization_solver, method)(*args, **kwargs)
    forward.__name__ = method
    forward.__doc__ = "OptimizationSolver." + method + " of the plugin this solver was built with"
    return forward


# what the transcription layer adds nothing to: guesses, the solve and its results, named costs / constraints
for _name in ("set_initial_guess", "get_initial_guess", "solve", "get_values", "get_cost_value", "get_cost_values", "get_constraint_multipliers",
              "add_cost", "add_constraint", "cost_function", "get_cost_expressions", "get_constraint_expressions"):
    setattr(MultipleShootingSolver, _name, _hand_through(_name))
del _name


class OptimalControlProblemInstance:
    """what `OptimalControlProblem.create` returns: (problem, all_variables, symbolic_structure), also by unpacking"""

    def __init__(self, problem, all_variables, symbolic_structure):
        self.problem, self.all_variables, self.symbolic_structure = problem, all_variables, symbolic_structure

    def __iter__(self):
        return iter((self.problem, self.all_variables, self.symbolic_structure))


class OptimalControlProblem:
    """The user-facing problem (base/optimal_control_problem.py, base/problem.py:80-200): owns an optimal-control solver, forwards the
    description of the problem to it and turns its results into an `Output`."""

    def __init__(self, optimal_control_solver=None):
        if optimal_control_solver is None:
            raise ValueError("OptimalControlProblem needs an optimal_control_solver: MultipleShootingSolver(optimization_solver=...)")
        self._solver = optimal_control_solver
        self._output = None
        self._solver.register_problem(self)

    @classmethod
    def create(cls, input_structure, optimal_control_solver=None, **kwargs):
        problem = cls(optimal_control_solver=optimal_control_solver)
        problem._solver.generate_optimization_objects(input_structure=input_structure, **kwargs)
        return OptimalControlProblemInstance(problem, problem._solver.get_optimization_objects(), problem._solver.get_symbolic_structure())

    def solver(self):
        return self._solver

    # ---- description of the problem ---------------------------------------------------------------------------------------------
    def add_dynamics(self, dynamics, x0=None, t0=0.0, mode=None, name=None, x0_name=None, **kwargs):
        self._solver.add_dynamics(dynamics, x0=x0, t0=t0, mode=mode, name=name, x0_name=x0_name, **kwargs)

    def add_expression_to_horizon(self, expression, mode=None, apply_to_first_elements=False, name=None, **kwargs):
        self._solver.add_expression_to_horizon(expression, mode=mode, apply_to_first_elements=apply_to_first_elements, name=name, **kwargs)

    def add_cost(self, expression, scaling=1.0, name=None, **_):
        from .affine import as_cost
        if not _accepts_expressions(self._solver.get_optimization_solver()):
            raise TypedProblemError("add_cost")
        for label, cost in as_cost(expression, scaling, name):
            self._solver.add_cost(cost, name=label)

    def add_constraint(self, expression, expected_value=0.0, name=None, **_):
        from .affine import as_constraint
        if not _accepts_expressions(self._solver.get_optimization_solver()):
            raise TypedProblemError("add_constraint")
        for label, constraint in as_constraint(expression, expected_value, name):
            self._solver.add_constraint(constraint, name=label)

    def add_expression(self, mode, expression, name=None, **kwargs):
        from .problem import ExpressionType
        if mode == ExpressionType.subject_to:
            self.add_constraint(expression, name=name, **kwargs)
        elif mode == ExpressionType.minimize:
            self.add_cost(expression, name=name, **kwargs)

    def initial(self, variable):
        return self._solver.initial(variable)

    def final(self, variable):
        return self._solver.final(variable)

    # ---- guesses, solve, results -------------------------------------------------------------------------------------------------
    def set_initial_guess(self, initial_guess):
        self._solver.set_initial_guess(initial_guess)

    def get_initial_guess(self):
        return self._solver.get_initial_guess()

    def get_cost_expressions(self):
        return self._solver.get_cost_expressions()

    def get_constraint_expressions(self):
        return self._solver.get_constraint_expressions()

    def solve(self):
        self._solver.solve()
        self._output = Output(values=self._solver.get_values(), cost_value=self._solver.get_cost_value(),
                              cost_values=self._solver.get_cost_values(), constraint_multipliers=self._solver.get_constraint_multipliers())
        return self._output

    def get_output(self):
        if self._output is None:
            raise ProblemNotSolvedException
        return self._output
