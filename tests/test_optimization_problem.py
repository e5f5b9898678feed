"""The closed forms the reference's own test/test_optimization_problem.py holds for `OptimizationProblem`
(/root/reference/src/hippopt/base/optimization_problem.py:41-66 on base/problem.py:80-200), restated on this build's facade with the
CPU plugin that takes expressions (`AffineSolver`: affine and quadratic forms, SciPy trust-constr) — SURVEY §8c pins (i):

    box-constrained separable QP: optimum and cost in closed form         test_optimization_problem.py:30-68
    the same with the bounds as parameters                                :80-126
    the same over a LIST input structure                                  :129-181
    a term switched between cost and constraint                           :194-254
    an infeasible problem raises the plugin's failure                     :257-267
    ... unless the iterate callback saved an intermediate solution        :270-283

No CasADi, no GPU.  What is NOT restated: `OptiSolver.to_function` (:286-330), which returns a CasADi Function."""
import dataclasses

import numpy as np
import pytest

import hippopt_amd as hp
from hippopt_amd.base import opti_callback
from hippopt_amd.base.affine import power
from hippopt_amd.base.affine_solver import AffineFailure, AffineSolver
from hippopt_amd.base.optimization_object import OptimizationObject, StorageType, default_storage_field


@dataclasses.dataclass
class MyTestVar(OptimizationObject):
    variable: StorageType = default_storage_field(hp.Variable)
    size: dataclasses.InitVar[int] = dataclasses.field(default=3)

    def __post_init__(self, size: int = 3):
        self.variable = np.zeros(size)


@dataclasses.dataclass
class MyTestVarAndPar(OptimizationObject):
    composite: MyTestVar = dataclasses.field(default_factory=MyTestVar)
    parameter: StorageType = default_storage_field(hp.Parameter)

    def __post_init__(self):
        self.parameter = np.zeros(3)


@dataclasses.dataclass
class SwitchVar(OptimizationObject):
    x: StorageType = default_storage_field(hp.Variable)
    y: StorageType = default_storage_field(hp.Variable)

    def __post_init__(self):
        self.x = np.zeros(1)
        self.y = np.zeros(1)


def box_qp(a, b, c):
    """min sum a x^2 + b x  s.t. x >= c, entry by entry: x* = max(-b / 2a, c) and its cost (the reference test's own closed form)"""
    free = -b / (2 * a)
    x = np.where(free >= c, free, c)
    cost = np.where(free >= c, -b ** 2 / (4 * a), a * c ** 2 + b * c)
    return x, float(cost.sum())


def coefficients(size):
    np.random.seed(123)
    return 10.0 * np.random.rand(size) + 0.01, 20.0 * np.random.rand(size) - 10.0, 20.0 * np.random.rand(size) - 10.0


def test_create_returns_problem_and_variables_and_defaults_to_the_expression_plugin():
    instance = hp.OptimizationProblem.create(input_structure=MyTestVar(size=2))
    problem, var = instance
    assert instance.problem is problem and instance.variables is var
    assert isinstance(problem.solver(), AffineSolver) and problem.solver().get_problem() is problem
    assert problem.solver().get_optimization_objects() is var and len(var.variable) == 2
    with pytest.raises(hp.base.ProblemNotSolvedException):
        problem.get_output()
    # anything that is not an OptimizationSolver -> the default plugin (optimization_problem.py:45-49)
    other, _ = hp.OptimizationProblem.create(input_structure=MyTestVar(), optimization_solver="ipopt")
    assert isinstance(other.solver(), AffineSolver)


def test_box_constrained_qp_reaches_its_closed_form():
    size = 4
    problem, var = hp.OptimizationProblem.create(input_structure=MyTestVar(size=size))
    a, b, c = coefficients(size)
    problem.add_expression(mode=hp.ExpressionType.minimize,
                           expression=(a[k] * power(var.variable[k], 2) + b[k] * var.variable[k] for k in range(size)))
    problem.add_expression(mode=hp.ExpressionType.subject_to, expression=(var.variable[k] >= c[k] for k in range(size)))
    output = problem.solve()
    expected_x, expected_cost = box_qp(a, b, c)
    assert output.values.variable.reshape(-1) == pytest.approx(expected_x, abs=1e-6)
    assert output.cost_value == pytest.approx(expected_cost)
    assert problem.solver().get_values().variable.reshape(-1) == pytest.approx(expected_x, abs=1e-6)
    assert problem.solver().get_cost_value() == pytest.approx(expected_cost)
    assert problem.get_output() is output
    # the generator was unrolled: one unnamed cost / constraint per entry, each with its own value / multiplier
    assert len(output.cost_values) == size and len(output.constraint_multipliers) == size
    assert sum(output.cost_values.values()) == pytest.approx(expected_cost)
    active = -b / (2 * a) < c
    lam = np.array([float(np.ravel(v)[0]) for v in output.constraint_multipliers.values()])
    assert np.all(np.abs(lam[~active]) < 1e-6) and np.all(np.abs(lam[active]) > 1e-6)


def test_box_constrained_qp_with_the_bounds_as_parameters():
    problem, var = hp.OptimizationProblem.create(input_structure=MyTestVarAndPar())
    initial_guess = MyTestVarAndPar()
    a, b, c = coefficients(3)
    initial_guess.parameter = c
    problem.add_expression(mode=hp.ExpressionType.minimize,
                           expression=(a[k] * power(var.composite.variable[k], 2) + b[k] * var.composite.variable[k] for k in range(3)))
    problem.add_expression(mode=hp.ExpressionType.subject_to, expression=(var.composite.variable[k] >= var.parameter[k] for k in range(3)))
    problem.solver().set_initial_guess(initial_guess=initial_guess)
    output = problem.solve()
    expected_x, expected_cost = box_qp(a, b, c)
    assert output.values.composite.variable.reshape(-1) == pytest.approx(expected_x, abs=1e-6)
    assert output.cost_value == pytest.approx(expected_cost)
    assert output.values.parameter.reshape(-1) == pytest.approx(c)
    assert problem.solver().get_values().composite.variable.reshape(-1) == pytest.approx(expected_x, abs=1e-6)
    assert problem.solver().get_cost_value() == pytest.approx(expected_cost)
    # a parameter without a value is refused before anything is solved (opti_solver.py:447-450)
    fresh, v2 = hp.OptimizationProblem.create(input_structure=MyTestVarAndPar())
    fresh.add_cost(power(v2.composite.variable[0], 2))
    guess = MyTestVarAndPar()
    guess.parameter = None
    fresh.solver()._guess.parameter = None
    with pytest.raises(ValueError, match="parameter"):
        fresh.solve()


def test_box_constrained_qp_over_a_list_of_objects():
    initial_guess = [MyTestVarAndPar() for _ in range(3)]
    problem, var = hp.OptimizationProblem.create(input_structure=initial_guess)
    np.random.seed(123)
    a, b, c = [], [], []
    for j in range(len(initial_guess)):
        a.append(10.0 * np.random.rand(3) + 0.01)
        b.append(20.0 * np.random.rand(3) - 10.0)
        c.append(20.0 * np.random.rand(3) - 10.0)
        initial_guess[j].parameter = c[j]
    problem.add_cost(a[j][k] * power(var[j].composite.variable[k], 2) + b[j][k] * var[j].composite.variable[k]
                     for j in range(len(initial_guess)) for k in range(3))
    problem.add_constraint(var[j].composite.variable[k] >= c[j][k] for j in range(len(initial_guess)) for k in range(3))
    problem.solver().set_initial_guess(initial_guess=initial_guess)
    output = problem.solve()
    expected_cost = 0.0
    for i in range(len(initial_guess)):
        expected_x, cost = box_qp(a[i], b[i], c[i])
        expected_cost += cost
        assert output.values[i].composite.variable.reshape(-1) == pytest.approx(expected_x, abs=1e-6)
        assert output.values[i].parameter.reshape(-1) == pytest.approx(c[i])
    assert output.cost_value == pytest.approx(expected_cost)
    assert problem.solver().get_cost_value() == pytest.approx(expected_cost)


def test_a_term_switched_from_cost_to_constraint():
    initial_problem, variables = hp.OptimizationProblem.create(input_structure=SwitchVar())
    a = 10
    initial_problem.add_expression(hp.ExpressionType.minimize, variables.x * variables.x)
    initial_problem.add_expression(hp.ExpressionType.minimize, a * variables.y * variables.y)
    initial_problem.add_expression(hp.ExpressionType.subject_to, variables.x + variables.y == a - 1)
    output = initial_problem.solve()
    # min x^2 + a y^2 on x + y = a - 1: x = a (a - 1) / (a + 1), cost a (a - 1)^2 / (a + 1) — what the reference test brackets within 10 %
    assert output.cost_value == pytest.approx(a * (a - 1) ** 2 / (a + 1), rel=1e-8)
    assert output.cost_value == pytest.approx(expected=a + (a - 2) ** 2, rel=0.1)
    assert float(np.ravel(output.values.x)[0]) == pytest.approx(a - 2, rel=0.1)

    new_problem, new_variables = hp.OptimizationProblem.create(input_structure=SwitchVar())
    new_problem.add_expression(hp.ExpressionType.minimize, a * new_variables.y * new_variables.y)
    new_problem.add_expression(hp.ExpressionType.subject_to, new_variables.x + new_variables.y == a - 1)
    new_problem.add_expression(hp.ExpressionType.subject_to, new_variables.x * new_variables.x + 1, expected_value=1)   # a bare scalar: == expected_value
    output = new_problem.solve()
    assert output.cost_value == pytest.approx(expected=a * (a - 1) ** 2, rel=0.1)
    assert float(np.ravel(output.values.x)[0]) == pytest.approx(0, abs=1e-4)


def test_a_term_switched_from_constraint_to_cost():
    initial_problem, variables = hp.OptimizationProblem.create(input_structure=SwitchVar())
    a = 10
    initial_problem.add_expression(hp.ExpressionType.minimize, (variables.x - 5) ** 2)
    initial_problem.add_expression(hp.ExpressionType.minimize, a * variables.y * variables.y)
    initial_problem.add_expression(hp.ExpressionType.subject_to, variables.x + variables.y == a - 1)
    initial_output = initial_problem.solve()

    new_problem, new_variables = hp.OptimizationProblem.create(input_structure=SwitchVar())
    new_problem.add_expression(hp.ExpressionType.minimize, a * new_variables.y * new_variables.y)
    new_problem.add_expression(hp.ExpressionType.subject_to, new_variables.x + new_variables.y == a - 1)
    new_problem.add_expression(hp.ExpressionType.minimize, new_variables.x == 5, scaling=1.0, name="new_cost")   # an equality as a cost: sumsqr of its difference
    output = new_problem.solve()
    assert output.cost_value == pytest.approx(expected=initial_output.cost_value, rel=0.1)
    assert float(np.ravel(output.values.x)[0]) == pytest.approx(float(np.ravel(initial_output.values.x)[0]), abs=1e-6)
    assert output.cost_values["new_cost"] == pytest.approx((float(np.ravel(output.values.x)[0]) - 5) ** 2)
    # an inequality has no cost form (base/problem.py:111-114)
    with pytest.raises(ValueError, match="inequality"):
        new_problem.add_expression(hp.ExpressionType.minimize, new_variables.x <= 5)
    # ExpressionType.skip adds nothing
    before = (len(new_problem.get_cost_expressions()), len(new_problem.get_constraint_expressions()))
    new_problem.add_expression(hp.ExpressionType.skip, new_variables.x == 7)
    assert before == (len(new_problem.get_cost_expressions()), len(new_problem.get_constraint_expressions()))


def test_an_infeasible_problem_raises_the_plugins_failure():
    problem, variables = hp.OptimizationProblem.create(input_structure=SwitchVar())
    problem.add_constraint(variables.x == 1)
    problem.add_constraint(variables.x == 10)
    with pytest.raises(AffineFailure) as err:
        problem.solve()
    print("Received error: ", err.value)


def test_the_iterate_callback_saves_a_failed_solve():
    solver = AffineSolver(callback_criterion=opti_callback.BestCost() | opti_callback.BestPrimalInfeasibility())
    problem, variables = hp.OptimizationProblem.create(input_structure=SwitchVar(), optimization_solver=solver)
    assert problem.solver() is solver
    problem.add_constraint(variables.x <= 1)
    problem.add_constraint(variables.x >= 0)
    problem.add_constraint(variables.x ** 2 == 10)
    output = problem.solve()          # infeasible — and no exception: the best iterate the callback saw is handed back (opti_solver.py:479-520)
    assert solver._solve_info["used_saved_iterate"] and solver._solve_info["failure"]
    assert np.isfinite(float(np.ravel(output.values.x)[0]))
    # the same problem without a criterion raises
    plain, v = hp.OptimizationProblem.create(input_structure=SwitchVar())
    plain.add_constraint(v.x <= 1)
    plain.add_constraint(v.x >= 0)
    plain.add_constraint(v.x ** 2 == 10)
    with pytest.raises(AffineFailure):
        plain.solve()


def test_a_typed_plugin_refuses_expressions_through_this_facade_too(model):
    """the engine-backed plugin's problem is the typed list built into the engine: create() works (structure, guesses), add_cost does not"""
    from hippopt_amd.hipnlp_solver import HipNlpSolver
    from hippopt_amd.turnkey_planners.humanoid_pose_finder import Planner, Settings
    planner = Planner(Settings(joints_name_list=["j%d" % i for i in range(23)]), model)      # (no device needed until solve())
    assert isinstance(planner.op, hp.base.OptimizationProblemInstance) and planner.op.problem.solver() is planner.optimization_solver
    assert isinstance(planner.optimization_solver, HipNlpSolver) and planner.op.variables is planner.optimization_solver.get_optimization_objects()
    with pytest.raises(hp.TypedProblemError):
        planner.op.problem.add_cost(1.0)
