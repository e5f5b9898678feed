"""The order in which jac g reaches the NLP driver must be invisible to a solve.

The reference hands IPOPT CasADi's CCS triplet order (/root/reference/src/hippopt/base/opti_solver.py:444-479 -> nlpsol; the scripts
main_single_step_flat_ground.py / main_periodic_step.py:109-134 set the options).  HipNlpSolver creates varying-first handles by default
(the host path then moves the varying entries of a knot block as one run and leaves the constants where they are).  north_star's one
solver-level requirement is an identical iterate sequence: here the kinodynamic planner is solved twice — single step N = 30 and
periodic N = 10, a few dozen iterations, through the product's own solver path — once per order, and every iterate, the outputs and the
named multipliers are compared.  The driver is SciPy trust-constr (IPOPT / MUMPS are not in the image: what cannot be verified here is
said in INTEGRATION.md); it builds its sparse matrix from the (row, column, value) triplets, as IPOPT does."""
import numpy as np
import pytest

from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings
from hippopt_amd.synthetic import make_workload

pytestmark = pytest.mark.gpu


def solve(model, numeric, vary_first, seed, max_iter, devices=None):
    from hippopt_amd.turnkey_planners.humanoid_kinodynamic import Planner, Settings
    st = Settings.from_numeric(numeric, solver_options={"max_iter": max_iter})
    pl = Planner(st, model, error_on_fail=False, inner_solver="trust-constr", devices=devices)
    sol = pl.optimization_solver
    sol._jac_varying_first = vary_first            # (before the engine exists: HipNlpSolver(..., jac_varying_first=) through the planner's wiring)
    x, p = make_workload(st, model, batch=1, seed=seed)
    guess = pl.get_initial_guess()
    guess.from_dict({n: x[0][off:off + size].reshape(shape) for n, (off, size, shape) in sol._var_index.items()})
    guess.from_dict({n: p[0][off:off + size].reshape(shape) for n, (off, size, shape) in sol._par_index.items()})
    sol.set_initial_guess(guess)
    sol.iterate_trace = []
    out = pl.solve()
    eng = sol.engine()
    assert eng.jac_varying_first == vary_first and eng.lifted
    ir, jc = sol.nlp_view().sparsity()
    return out, sol.iterate_trace, list(zip(ir.tolist(), jc.tolist())), sol._last_info


# (iterations: the stand-in driver factorises the augmented system with SuperLU — 1.3 s per iteration on the single step N = 30, 8 s on the
#  periodic N = 10, whose first-to-last coupling fills the factors: 25 and 7 iterations keep the two cases at a minute each)
@pytest.mark.parametrize("maker,horizon,seed,iterations", [(single_step_settings, 30, 61, 25), (periodic_step_settings, 10, 62, 7)])
def test_iterate_sequence_does_not_depend_on_the_triplet_order(model, maker, horizon, seed, iterations):
    numeric = maker(horizon, model)
    runs = {vf: solve(model, numeric, vf, seed, iterations) for vf in (True, False)}
    (out_v, trace_v, rc_v, info_v), (out_c, trace_c, rc_c, info_c) = runs[True], runs[False]
    assert rc_v != rc_c and sorted(rc_v) == sorted(rc_c)                 # two orders of one pattern
    assert rc_c == sorted(rc_c, key=lambda t: (t[1], t[0]))              # the reference's: column major (CCS)
    assert len(trace_v) == len(trace_c) >= iterations - 1
    worst = 0.0
    for (it_v, x_v, f_v, pr_v), (it_c, x_c, f_c, pr_c) in zip(trace_v, trace_c):
        assert it_v == it_c
        worst = max(worst, float(np.max(np.abs(x_v - x_c))))
        assert abs(f_v - f_c) <= 1e-12 * max(1.0, abs(f_c)) and abs(pr_v - pr_c) <= 1e-12 * max(1.0, abs(pr_c))
    assert worst <= 1e-12, worst
    assert info_v["iterations"] == info_c["iterations"] and info_v["callbacks"] == info_c["callbacks"]
    # Output: values, cost, per-cost values, named multipliers
    assert abs(out_v.cost_value - out_c.cost_value) <= 1e-12 * max(1.0, abs(out_c.cost_value))
    vals_v, vals_c = out_v.values.to_dict(), out_c.values.to_dict()
    assert vals_v.keys() == vals_c.keys()
    for name in vals_v:
        if vals_v[name] is not None:
            assert np.max(np.abs(np.asarray(vals_v[name], float) - np.asarray(vals_c[name], float))) <= 1e-12, name
    assert out_v.cost_values.keys() == out_c.cost_values.keys()
    for name in out_v.cost_values:
        assert abs(out_v.cost_values[name] - out_c.cost_values[name]) <= 1e-12 * max(1.0, abs(out_c.cost_values[name])), name
    assert out_v.constraint_multipliers.keys() == out_c.constraint_multipliers.keys()
    for name in out_v.constraint_multipliers:
        a, b = np.asarray(out_v.constraint_multipliers[name]), np.asarray(out_c.constraint_multipliers[name])
        assert np.max(np.abs(a - b)) <= 1e-10 * max(1.0, float(np.max(np.abs(b)))), name
    print("iterates compared: %d, largest |x_varying_first - x_ccs| over all of them: %.3g" % (len(trace_v), worst))
