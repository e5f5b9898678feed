#!/usr/bin/env python3
"""GPU box: hipnlp_eval_hess through host buffers (value array registered by the handle: direct kernel stores), two handles in one
process alternating — every entry stored at the end of a knot's program (HIPNLP_EARLY_STORE=0 on the diagnostic build) against the run at
the start of a knot's block leaving behind its barrier (the default).  HESS_N, HESS_WORKLOAD=periodic|stairs."""
import os
# (the environment overrides below exist in the diagnostic build of the library only: __graft_entry__.build() -> tests/_build)
DIAG_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "_build", "libhipnlp_diag.so")
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
for tag, env in (("at the end", "0"), ("early run", "1")):
    os.environ["HIPNLP_EARLY_STORE"] = env
    engs[tag] = HipNlp(st, model, library=DIAG_SO)
    engs[tag].set_params(p)
del os.environ["HIPNLP_EARLY_STORE"]
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
    print("%-14s %.1f us per hipnlp_eval_hess (N = %d, best of 5 x 100 calls)" % (t, 1e6 * best[t], N), engs[t].host_stats())
