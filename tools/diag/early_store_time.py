#!/usr/bin/env python3
"""GPU box: the host-buffer callback (all four outputs, varying-first handle, caller arrays registered by the handle) with and without the
early copy-out (HIPNLP_EARLY_STORE, read by hipnlp_create): wall clock per call, the library's own breakdown, the kernel's duration by events."""
import os
# (the environment overrides below exist in the diagnostic build of the library only: __graft_entry__.build() -> tests/_build)
DIAG_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "_build", "libhipnlp_diag.so")
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hippopt_amd.hipnlp import HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402

model = synthetic_ergocub()
st = periodic_step_settings(100, model)
x, p = make_workload(st, model, batch=1, seed=1)
xs = [x + 1e-3 * i for i in range(4)]
engs = {}
for e in ("1", "0"):
    os.environ["HIPNLP_EARLY_STORE"] = e
    engs[e] = HipNlp(st, model, jac_varying_first=True, library=DIAG_SO)
    engs[e].set_params(p)
    engs[e].set_prefetch(())
del os.environ["HIPNLP_EARLY_STORE"]
outs = {e: engs[e].eval(x) for e in engs}
for rep in range(3):
    for e in ("1", "0"):
        eng, out = engs[e], outs[e]
        for i in range(20):
            eng.eval(xs[i % 4], out=out)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(200):
                eng.eval(xs[i % 4], out=out)
            best = min(best, (time.perf_counter() - t0) / 200)
        eng.set_host_timing(True)
        ks = []
        for i in range(50):
            eng.eval(xs[i % 4], out=out)
            ks.append(eng.last_kernel_ms())
        eng.set_host_timing(False)
        print("early=%s  %.2f us per call  library %s  kernel by events %.2f us (median of 50)" % (e, 1e6 * best, [round(float(v), 2) for v in eng.host_breakdown()], 1e3 * float(np.median(ks))), flush=True)
assert all(np.array_equal(a, b) for a, b in zip(outs["1"], outs["0"]))
