#!/usr/bin/env python3
"""Diagnostic (GPU box): the kinodynamic planner mirror solved with the exact Hessian of the Lagrangian (hipnlp_eval_hess) and with
the quasi-Newton approximation (what `hessian_approximation = limited-memory` means for the stand-in driver), same start, same
driver (SciPy trust-constr; IPOPT is not in the image).  Prints iterations, constraint violation and wall-clock per mode."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hippopt_amd.kinodyn_settings import single_step_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402
from hippopt_amd.turnkey_planners.humanoid_kinodynamic import Planner, Settings  # noqa: E402

N = int(os.environ.get("SOLVE_N", "8"))
ITERS = int(os.environ.get("SOLVE_ITERS", "150"))
model = synthetic_ergocub()
for mode in ("exact", "limited-memory"):
    st = Settings.from_numeric(single_step_settings(N, model), solver_options={"max_iter": ITERS, "hessian_approximation": mode})
    pl = Planner(st, model, error_on_fail=False)
    x, p = make_workload(st, model, batch=1, seed=8)
    guess = pl.get_initial_guess()
    names = pl.optimization_solver._var_index
    guess.from_dict({n: x[0][off:off + size].reshape(shape) for n, (off, size, shape) in names.items()})
    pars = pl.optimization_solver._par_index
    guess.from_dict({n: p[0][off:off + size].reshape(shape) for n, (off, size, shape) in pars.items()})
    pl.optimization_solver.set_initial_guess(guess)
    eng = pl.optimization_solver.engine()
    eng.set_params(p)
    _, _, g0, _ = eng.eval(x)
    _, _, lbg, ubg = eng.bounds()
    viol0 = np.max(np.maximum(0, np.maximum(lbg - g0[0], g0[0] - ubg)))
    t0 = time.perf_counter()
    out = pl.solve()
    dt = time.perf_counter() - t0
    info = pl.optimization_solver._last_info
    print("N=%d %-15s iterations %4d  cost %.6f  constraint violation %.3e (start %.3e)  status %s  (%.2f s)" % (
        N, mode, info.get("iterations", -1), out.cost_value, info.get("constr_violation", float("nan")), viol0, info.get("status"), dt), flush=True)
