"""`OptimizationProblem` — the facade of a problem WITHOUT a horizon (reference: base/optimization_problem.py:16-66, on top of
base/problem.py:80-200).  `OptimizationProblem.create(input_structure, optimization_solver)` asks the solver plugin to generate the
optimisation objects for the structure and returns `(problem, variables)`; the problem forwards `add_cost` / `add_constraint` /
`add_expression` to the plugin with the conversions of base/problem.py:95-174 (an equality as a cost is the sum of squares of its
difference, a bare scalar constraint is `expression == expected_value`, generators are unrolled with `{i}` behind the name) and
turns the plugin's results into an `Output`.

It is the injection point the reference's pose finder uses (turnkey_planners/humanoid_pose_finder/planner.py:334-343:
`hp.OptimizationProblem.create(input_structure=self.variables, optimization_solver=self.optimization_solver)`); the mirror of that
planner goes through it with the engine-backed `HipNlpSolver(problem="pose")`.

The reference's default plugin is CasADi's `OptiSolver()`; this build's default is `AffineSolver()` — the CPU plugin that takes
expressions (affine and quadratic forms: the closed-form problems of the reference's own test/test_optimization_problem.py).  A plugin
whose problem is typed (`HipNlpSolver`) refuses `add_cost` / `add_constraint` with a message that says why."""
from .optimal_control import OptimizationSolver, TypedProblemError, _accepts_expressions
from .problem import ExpressionType, Output, ProblemNotSolvedException


class OptimizationProblemInstance:
    """what `OptimizationProblem.create` returns: (problem, variables), also by unpacking (optimization_problem.py:11-33)"""

    def __init__(self, problem, variables):
        self.problem, self.variables = problem, variables

    def __iter__(self):
        return iter((self.problem, self.variables))


class OptimizationProblem:
    def __init__(self, optimization_solver=None):
        if optimization_solver is None or not isinstance(optimization_solver, OptimizationSolver):   # (optimization_problem.py:45-49: anything else -> the default plugin)
            from .affine_solver import AffineSolver
            optimization_solver = AffineSolver()
        self._solver = optimization_solver
        self._output = None
        self._solver.register_problem(self)

    @classmethod
    def create(cls, input_structure, optimization_solver=None, **kwargs):
        new_problem = cls(optimization_solver=optimization_solver)
        new_problem._solver.generate_optimization_objects(input_structure=input_structure, **kwargs)
        return OptimizationProblemInstance(new_problem, new_problem._solver.get_optimization_objects())

    def solver(self):
        return self._solver

    # ---- guesses ----------------------------------------------------------------------------------------------------------------
    def set_initial_guess(self, initial_guess):
        self._solver.set_initial_guess(initial_guess)

    def get_initial_guess(self):
        return self._solver.get_initial_guess()

    # ---- description of the problem (base/problem.py:95-190) ------------------------------------------------------------------------
    def add_cost(self, expression, scaling=1.0, name=None, **_):
        from .affine import as_cost
        if not _accepts_expressions(self._solver):
            raise TypedProblemError("add_cost")
        for label, cost in as_cost(expression, scaling, name):
            self._solver.add_cost(cost, name=label)

    def add_constraint(self, expression, expected_value=0.0, name=None, **_):
        from .affine import as_constraint
        if not _accepts_expressions(self._solver):
            raise TypedProblemError("add_constraint")
        for label, constraint in as_constraint(expression, expected_value, name):
            self._solver.add_constraint(constraint, name=label)

    def add_expression(self, mode, expression, name=None, **kwargs):
        if mode == ExpressionType.subject_to:
            self.add_constraint(expression=expression, name=name, **kwargs)
        elif mode == ExpressionType.minimize:
            self.add_cost(expression=expression, name=name, **kwargs)

    def get_cost_expressions(self):
        return self._solver.get_cost_expressions()

    def get_constraint_expressions(self):
        return self._solver.get_constraint_expressions()

    # ---- solve, results ----------------------------------------------------------------------------------------------------------
    def solve(self):
        self._solver.solve()
        self._output = Output(values=self._solver.get_values(), cost_value=self._solver.get_cost_value(),
                              cost_values=self._solver.get_cost_values(), constraint_multipliers=self._solver.get_constraint_multipliers())
        return self._output

    def get_output(self):
        if self._output is None:
            raise ProblemNotSolvedException
        return self._output
