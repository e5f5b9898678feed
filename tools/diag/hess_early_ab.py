#!/usr/bin/env python3
"""GPU box: hipnlp_eval_hess through host buffers (value array registered by the handle: direct kernel stores), two handles in one
process alternating — every entry stored at the end of a knot's program against the run at the start of a knot's block leaving behind its
barrier (hipnlp_set_hessian_early_run 0 / 1), and what a third handle left to decide by itself chose.  HESS_N, HESS_WORKLOAD=periodic|stairs."""
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
engs = {}
for tag, mode in (("at the end", False), ("early run", True), ("decided by handle", None)):
    engs[tag] = HipNlp(st, model)
    engs[tag].set_hessian_early_run(mode)
    engs[tag].set_params(p)
lam = np.random.RandomState(0).standard_normal((1, engs["early run"].m))
xs = [x + 1e-4 * i for i in range(4)]
outs = {t: e.eval_hess(x, 1.0, lam).copy() for t, e in engs.items()}
assert np.array_equal(outs["at the end"], outs["early run"])
for t, e in engs.items():
    for i in range(20):
        e.eval_hess(xs[i % 4], 1.0, lam, out=outs[t])
best = {t: 1e9 for t in engs}
for rep in range(5):
    for t, e in engs.items():
        t0 = time.perf_counter()
        for i in range(100):
            e.eval_hess(xs[i % 4], 1.0, lam, out=outs[t])
        best[t] = min(best[t], (time.perf_counter() - t0) / 100)
for t in engs:
    print("%-18s %.1f us per hipnlp_eval_hess (N = %d, best of 5 x 100 calls)  %s" % (t, 1e6 * best[t], N, engs[t].hessian_early_run()))
