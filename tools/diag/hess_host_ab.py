#!/usr/bin/env python3
"""GPU box: hipnlp_eval_hess through host buffers (value array registered by the handle: direct kernel stores), two handles in one
process alternating — multipliers through the H2D copy command (default) against the kernel reading them from pinned memory
(HIPNLP_HESS_LAM_ZERO_COPY=1)."""
import os
# (the environment overrides below exist in the diagnostic build of the library only: __graft_entry__.build() -> tests/_build)
DIAG_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "_build", "libhipnlp_diag.so")
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hippopt_amd.hipnlp import HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402

N = int(os.environ.get("HESS_N", "100"))
model = synthetic_ergocub()
st = periodic_step_settings(N, model)
x, p = make_workload(st, model, batch=1, seed=3)
engs = {}
for tag, env in (("copy command", "0"), ("zero copy", "1")):
    os.environ["HIPNLP_HESS_LAM_ZERO_COPY"] = env
    engs[tag] = HipNlp(st, model, library=DIAG_SO)
    engs[tag].set_params(p)
del os.environ["HIPNLP_HESS_LAM_ZERO_COPY"]
lam = np.random.RandomState(0).standard_normal((1, engs["zero copy"].m))
xs = [x + 1e-4 * i for i in range(4)]
outs = {t: e.eval_hess(x, 1.0, lam).copy() for t, e in engs.items()}
assert np.array_equal(outs["copy command"], outs["zero copy"])
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
