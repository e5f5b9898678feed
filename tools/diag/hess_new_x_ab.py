#!/usr/bin/env python3
"""GPU box: hipnlp_eval_hess_at in IPOPT's order at an accepted iterate — the callbacks at x, then eval_h — with new_x = TRUE (x copied into
the staging block again) against new_x = FALSE (the copy the callbacks made is used).  One handle, the two flags alternating pass by pass;
only the Hessian calls are timed.  Also: the plain loop of Hessian calls at a new x each (what bench.py reports as host_visible_ms).
HESS_N, HESS_WORKLOAD=periodic|stairs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hippopt_amd.hipnlp import HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402

N = int(os.environ.get("HESS_N", "100"))
model = synthetic_ergocub()
st = (stairs_settings if os.environ.get("HESS_WORKLOAD") == "stairs" else periodic_step_settings)(N, model)
x, p = make_workload(st, model, batch=1, seed=3)
eng = HipNlp(st, model)
eng.set_params(p)
lam = np.random.RandomState(0).standard_normal((1, eng.m))
xs = [x + 1e-4 * i for i in range(4)]
out = eng.eval_hess(x, 1.0, lam).copy()
ref = [eng.eval_hess(xi, 1.0, lam).copy() for xi in xs]
for i in range(20):
    eng.eval_hess(xs[i % 4], 1.0, lam, out=out)
best = {"new_x = TRUE": 1e9, "new_x = FALSE": 1e9, "Hessian calls alone, new x each": 1e9}
for rep in range(6):
    for tag, flag in (("new_x = TRUE", True), ("new_x = FALSE", False)):
        acc = 0.0
        for i in range(100):
            eng.eval(xs[i % 4], want=("f",))
            t0 = time.perf_counter()
            eng.eval_hess(xs[i % 4], 1.0, lam, out=out, new_x=flag)
            acc += time.perf_counter() - t0
            if i < 4:
                assert np.array_equal(out, ref[i % 4]), (tag, i)
        best[tag] = min(best[tag], acc / 100)
    t0 = time.perf_counter()
    for i in range(100):
        eng.eval_hess(xs[i % 4], 1.0, lam, out=out)
    best["Hessian calls alone, new x each"] = min(best["Hessian calls alone, new x each"], (time.perf_counter() - t0) / 100)
for tag, v in best.items():
    print("%-34s %.1f us per Hessian call (N = %d, best of 6 passes of 100)" % (tag, 1e6 * v, N))
print(eng.host_stats())
