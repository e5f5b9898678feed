"""The cyipopt binding of HipNlpSolver (`_solve_ipopt`: objective / gradient / constraints / jacobian(structure) / hessian(structure) /
intermediate, `get_current_iterate`, multipliers incl. the bound multipliers of the lifted single-variable rows) executed against a
TEST-ONLY stand-in of cyipopt's API (tests/cyipopt_standin: cyipopt and IPOPT are not in the image) with the engine emulated on the
host — the path IPOPT would drive, end to end, without a GPU."""
import os
import sys

import numpy as np
import pytest

from hippopt_amd.base.opti_callback import AcceptablePrimalInfeasibility, BestCost
from hippopt_amd.kinodyn_settings import single_step_settings
from hippopt_amd.synthetic import make_workload


@pytest.fixture()
def cyipopt_standin(monkeypatch):
    monkeypatch.syspath_prepend(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cyipopt_standin"))
    sys.modules.pop("cyipopt", None)
    yield
    sys.modules.pop("cyipopt", None)


def test_ipopt_driver_end_to_end_on_the_emulated_engine(model, cyipopt_standin):
    from emu_engine import EmuEngine
    from hippopt_amd.turnkey_planners.humanoid_kinodynamic import Planner, Settings
    N = 3
    st = Settings.from_numeric(single_step_settings(N, model), solver_options={"max_iter": 6, "hessian_approximation": "limited-memory"},
                               use_opti_callback=True, acceptable_constraint_violation=np.inf)
    pl = Planner(st, model, inner_solver="auto", error_on_fail=False)
    sol = pl.optimization_solver
    emu = EmuEngine(st, model, detect_simple_bounds=True)   # the handle HipNlpSolver creates: the reduced NLP (HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS)
    sol.engine = lambda: emu                      # the engine handle, emulated on the host
    x, p = make_workload(st, model, batch=1, seed=8)
    guess = pl.get_initial_guess()
    guess.from_dict({n: x[0][off:off + size].reshape(shape) for n, (off, size, shape) in sol._var_index.items()})
    guess.from_dict({n: p[0][off:off + size].reshape(shape) for n, (off, size, shape) in sol._par_index.items()})
    sol.set_initial_guess(guess)
    calls = []
    inner = sol._iterate_callback
    sol._iterate_callback = lambda it, xk, cost, inf_pr, lam, lam_x=None: (calls.append((it, None if lam is None else len(lam), lam_x is not None)),
                                                                             inner(it, xk, cost, inf_pr, lam, lam_x))[1]
    out = pl.solve()
    info = sol._last_info
    assert "status_msg" in info and info["nlp"]["simple_bounds_lifted"] == 70 * 2 + 47 + 81      # the cyipopt path ran, on the reduced problem
    assert calls and all(n == info["nlp"]["m"] and has_x for _, n, has_x in calls)             # intermediate() saw mult_g and the bound multipliers
    assert set(info["callbacks"]) >= {"f", "grad", "g", "jac", "evaluations"} and emu.evaluations >= info["callbacks"]["evaluations"]
    assert out.constraint_multipliers["joint_velocity_bounds"].shape == (N, 23)                 # lifted rows got their multipliers back
    assert out.constraint_multipliers["joint_position_dynamics"].shape == (N - 1, 23)
    cb = sol._callback
    assert cb.best_iteration is not None and cb.best_constraint_multipliers.size == emu.m_full      # best-iterate store fed from intermediate()
    assert np.isfinite(out.cost_value)
