"""BASELINE config 1: the falling-mass OCP of the reference's test/test_multiple_shooting.py:253-353, solved through the planner-level
facade with the reference's names — `OptimalControlProblem.create(input_structure, optimal_control_solver, horizon)`, four styles of
`add_dynamics`, `add_expression_to_horizon`, `initial` / `final`, named costs and multipliers — on the CPU plugin `AffineSolver`
(affine expressions, SciPy QP; no CasADi, no GPU).  The assertions are the reference test's: the three masses follow the explicit
Euler roll-out x_{i+1} = x_i + dt v_i, v_{i+1} = v_i + dt g, and foo sits on its bounds."""
import numpy as np
import pytest

import hippopt_amd as hp
from hippopt_amd import integrators
from hippopt_amd.base.affine import dot, sumsqr
from hippopt_amd.base.affine_solver import AffineSolver
from hippopt_amd.base.schema import declare, leaf, series

MASSES, HORIZON, DT, X0, V0, GRAVITY = 3, 100, 0.01, 1.0, 0.0, -9.81


def falling(x, v, g):
    """what the reference's test wraps in a cs.Function "dynamics": inputs x, v, g -> outputs x_dot, v_dot"""
    return {"x_dot": v, "v_dot": g}


# the structure of the reference's test (a list of three point masses over time, gravity as a parameter, a free variable `foo`),
# declared through the build's table form
Mass = declare("MassFallingState", {"x": leaf(hp.Variable, lambda: np.zeros(1)), "v": leaf(hp.Variable, lambda: np.zeros(1))}, module=__name__)


def _fill(self):
    self.masses = [Mass() for _ in range(MASSES)]


Tree = declare("MassFallingTestVariables", {"masses": series(), "g": leaf(hp.Parameter, lambda: GRAVITY), "foo": leaf(hp.Variable, lambda: np.zeros((3, 1)))},
               setup=_fill, module=__name__)


def expected_rollout():
    """explicit Euler from (X0, V0) under GRAVITY: the closed form the reference test steps through knot by knot"""
    x, v = np.empty(HORIZON), np.empty(HORIZON)
    x[0], v[0] = X0, V0
    for i in range(1, HORIZON):
        x[i], v[i] = x[i - 1] + DT * v[i - 1], v[i - 1] + DT * GRAVITY
    return x, v


def test_falling_masses_follow_the_euler_rollout():
    problem, var, symbolic = hp.OptimalControlProblem.create(
        input_structure=Tree(), optimal_control_solver=hp.MultipleShootingSolver(optimization_solver=AffineSolver()), horizon=HORIZON)
    assert problem.initial(symbolic.g) is problem.final(symbolic.g)          # a constant has one symbol for the whole horizon
    euler = {"dt": DT, "integrator": integrators.ForwardEuler}
    # mass 0: dynamics as a function with an argument map; its initial state by two separate constraints (one of them named)
    problem.add_dynamics(dot(["masses[0].x", "masses[0].v"]) == (falling, {"masses[0].x": "x", "masses[0].v": "v"}), **euler)
    pinned = var.masses[0][0].x == X0
    problem.add_constraint(pinned, name="initial_position")
    problem.add_constraint(var.masses[0][0].v == V0)
    # mass 1: the same dynamics as a COST, x0 given by name
    problem.add_dynamics(dot(["masses[1].x", "masses[1].v"]) == (falling, {"masses[1].x": "x", "masses[1].v": "v"}),
                         x0={"masses[1].x": X0, "masses[1].v": V0}, mode=hp.ExpressionType.minimize, x0_name="initial_condition", **euler)
    # mass 2: plain expressions, one state at a time, x0 as a scalar and as a {symbol: value} map
    problem.add_dynamics(dot(symbolic.masses[2].x) == symbolic.masses[2].v, x0=X0, x0_name="initial_condition_simple_x", **euler)
    problem.add_dynamics(dot(symbolic.masses[2].v) == ["g"], x0={symbolic.masses[2].v: V0}, x0_name="initial_condition_simple_v", **euler)
    # foo: >= 5 on every knot but the first, 0 at the first, 6 at the last, its square minimised everywhere
    problem.add_expression_to_horizon(expression=(symbolic.foo >= 5), apply_to_first_elements=False)
    problem.add_constraint(expression=problem.initial(symbolic.foo) == 0)
    problem.add_constraint(expression=problem.final(symbolic.foo) == 6.0)
    problem.add_expression_to_horizon(expression=sumsqr(symbolic.foo), apply_to_first_elements=True, mode=hp.ExpressionType.minimize)

    guess = Tree()
    guess.masses = guess.foo = None                   # a guess that only carries the parameter
    problem.set_initial_guess(guess)
    sol = problem.solve()

    # names: what was named is found under its name, generated names carry the knot index
    assert problem.get_constraint_expressions()["initial_position"] is pinned and "initial_position" in sol.constraint_multipliers
    assert {"initial_condition{0}", "initial_condition{1}"} <= set(problem.get_cost_expressions())
    assert {"initial_condition_simple_x{0}", "initial_condition_simple_v{0}"} <= set(sol.constraint_multipliers)
    # values: every mass on the roll-out, foo on its bounds (what test/test_multiple_shooting.py:336-353 asserts knot by knot)
    x_want, v_want = expected_rollout()
    for mass in range(MASSES):
        got_x = np.array([np.asarray(sol.values.masses[mass][i].x).item() for i in range(HORIZON)])
        got_v = np.array([np.asarray(sol.values.masses[mass][i].v).item() for i in range(HORIZON)])
        assert np.abs(got_x - x_want).max() < 1e-6 and np.abs(got_v - v_want).max() < 1e-6, mass
    foo_want = np.full((HORIZON, 3), 5.0)
    foo_want[0], foo_want[-1] = 0.0, 6.0
    assert np.abs(np.array([np.asarray(f).reshape(-1) for f in sol.values.foo]) - foo_want).max() < 1e-5
    assert sol.cost_value == pytest.approx(3 * (25.0 * (HORIZON - 2) + 36.0), rel=1e-6)     # sumsqr(foo): the only cost left at the optimum
    assert sum(sol.cost_values.values()) == pytest.approx(sol.cost_value, rel=1e-9)


def test_engine_backed_solver_refuses_expressions_with_a_pointer_to_the_typed_route(model):
    """The same facade on the engine-backed solver: structure, guesses and names work without a GPU; describing the problem by
    expressions is refused with a message that names the typed route (no silent fallback to a CPU evaluation)."""
    from hippopt_amd.kinodyn_settings import single_step_settings
    from hippopt_amd.turnkey_planners.humanoid_kinodynamic import Settings
    from hippopt_amd.turnkey_planners.humanoid_kinodynamic.variables import Variables
    N = 3
    st = Settings.from_numeric(single_step_settings(N, model))
    solver = hp.MultipleShootingSolver(optimization_solver=hp.HipNlpSolver(st, model))
    problem, var, symbolic = hp.OptimalControlProblem.create(input_structure=Variables(settings=st, kin_dyn_object=model),
                                                             optimal_control_solver=solver, horizon=N)
    assert len(var.system) == N and isinstance(problem.solver(), hp.MultipleShootingSolver)
    assert np.asarray(problem.initial("system.com")).shape == (3, 1) and problem.final("dt") is not None
    for call in (lambda: problem.add_dynamics(dot("system.com") == ["system.centroidal_momentum"], dt=0.1),
                 lambda: problem.add_expression_to_horizon(sumsqr(0.0)),
                 lambda: problem.add_cost(sumsqr(1.0)), lambda: problem.add_constraint(sumsqr(1.0))):
        with pytest.raises(hp.TypedProblemError, match="typed row and cost directory"):
            call()
    with pytest.raises(ValueError, match="not found"):
        problem.initial("system.no_such_leaf")
    with pytest.raises(ValueError, match="needs an optimization_solver"):
        hp.MultipleShootingSolver()
