"""BASELINE config 1: the falling-mass OCP of the reference's test/test_multiple_shooting.py:253-353, solved through the planner-level
facade with the reference's names — `OptimalControlProblem.create(input_structure, optimal_control_solver, horizon)`, four styles of
`add_dynamics`, `add_expression_to_horizon`, `initial` / `final`, named costs and multipliers — on the CPU plugin `AffineSolver`
(affine expressions, SciPy QP; no CasADi, no GPU).  The assertions are the reference test's: the three masses follow the explicit
Euler roll-out x_{i+1} = x_i + dt v_i, v_{i+1} = v_i + dt g, and foo sits on its bounds."""
import dataclasses

import numpy as np
import pytest

import hippopt_amd as hp
from hippopt_amd import integrators
from hippopt_amd.base.affine import dot, sumsqr
from hippopt_amd.base.affine_solver import AffineSolver


@dataclasses.dataclass
class MassFallingState(hp.OptimizationObject):
    x: hp.StorageType = hp.default_storage_field(hp.Variable)
    v: hp.StorageType = hp.default_storage_field(hp.Variable)

    def __post_init__(self):
        self.x, self.v = np.zeros(1), np.zeros(1)

    @staticmethod
    def get_dynamics():
        def dynamics(x, v, g):      # the reference's cs.Function "dynamics": inputs x, v, g -> outputs x_dot, v_dot
            return {"x_dot": v, "v_dot": g}
        return dynamics


@dataclasses.dataclass
class MassFallingTestVariables(hp.OptimizationObject):
    masses: list = dataclasses.field(metadata=hp.time_varying_metadata(), default=None)
    g: hp.StorageType = hp.default_storage_field(hp.Parameter)
    foo: hp.StorageType = hp.default_storage_field(hp.Variable)

    def __post_init__(self):
        self.g = -9.81
        self.masses = [MassFallingState() for _ in range(3)]
        self.foo = np.zeros((3, 1))


def test_falling_masses_follow_the_euler_rollout():
    guess = MassFallingTestVariables()
    guess.masses = None
    guess.foo = None
    horizon, dt, initial_position, initial_velocity = 100, 0.01, 1.0, 0.0

    problem, var, symbolic = hp.OptimalControlProblem.create(
        input_structure=MassFallingTestVariables(), optimal_control_solver=hp.MultipleShootingSolver(optimization_solver=AffineSolver()),
        horizon=horizon)
    assert problem.initial(symbolic.g) is problem.final(symbolic.g)

    problem.add_dynamics(dot(["masses[0].x", "masses[0].v"]) == (MassFallingState.get_dynamics(), {"masses[0].x": "x", "masses[0].v": "v"}),
                         dt=dt, integrator=integrators.ForwardEuler)
    initial_position_constraint = var.masses[0][0].x == initial_position
    problem.add_constraint(initial_position_constraint, name="initial_position")
    problem.add_constraint(var.masses[0][0].v == initial_velocity)

    problem.add_dynamics(dot(["masses[1].x", "masses[1].v"]) == (MassFallingState.get_dynamics(), {"masses[1].x": "x", "masses[1].v": "v"}),
                         dt=dt, x0={"masses[1].x": initial_position, "masses[1].v": initial_velocity}, integrator=integrators.ForwardEuler,
                         mode=hp.ExpressionType.minimize, x0_name="initial_condition")
    problem.add_dynamics(dot(symbolic.masses[2].x) == symbolic.masses[2].v, dt=dt, x0=initial_position, integrator=integrators.ForwardEuler,
                         x0_name="initial_condition_simple_x")
    problem.add_dynamics(dot(symbolic.masses[2].v) == ["g"], dt=dt, x0={symbolic.masses[2].v: initial_velocity},
                         integrator=integrators.ForwardEuler, x0_name="initial_condition_simple_v")

    problem.add_expression_to_horizon(expression=(symbolic.foo >= 5), apply_to_first_elements=False)
    problem.add_constraint(expression=problem.initial(symbolic.foo) == 0)
    problem.add_constraint(expression=problem.final(symbolic.foo) == 6.0)
    problem.add_expression_to_horizon(expression=sumsqr(symbolic.foo), apply_to_first_elements=True, mode=hp.ExpressionType.minimize)

    problem.set_initial_guess(guess)
    sol = problem.solve()

    assert problem.get_constraint_expressions()["initial_position"] is initial_position_constraint
    assert "initial_condition{0}" in problem.get_cost_expressions() and "initial_condition{1}" in problem.get_cost_expressions()
    assert "initial_position" in sol.constraint_multipliers
    assert "initial_condition_simple_x{0}" in sol.constraint_multipliers and "initial_condition_simple_v{0}" in sol.constraint_multipliers

    expected_position, expected_velocity = initial_position, initial_velocity
    for i in range(horizon):
        for mass in range(3):
            assert np.asarray(sol.values.masses[mass][i].x).item() == pytest.approx(expected_position, abs=1e-6)
            assert np.asarray(sol.values.masses[mass][i].v).item() == pytest.approx(expected_velocity, abs=1e-6)
        want = 0.0 if i == 0 else (6.0 if i == horizon - 1 else 5.0)
        assert np.asarray(sol.values.foo[i]).reshape(-1) == pytest.approx(want, abs=1e-5)
        expected_position += dt * expected_velocity
        expected_velocity += dt * guess.g
    assert sol.cost_value == pytest.approx(3 * (25.0 * (horizon - 2) + 36.0), rel=1e-6)     # sumsqr(foo) over the horizon: the only cost left
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
