#!/usr/bin/env python3
"""GPU box: life-cycle stress of the auto-registration — arrays freed (and their addresses reused by the allocator) while the handle
still holds their registration, handles destroyed after their arrays, many handles in one process."""
import gc
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hippopt_amd.hipnlp import HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402

model = synthetic_ergocub()
st = periodic_step_settings(100, model)
x, p = make_workload(st, model, batch=1, seed=9)
ref = HipNlp(st, model)
ref.set_params(p)
ref.set_auto_register(False)
want = ref.eval(x)
fallbacks = registered = 0
for rep in range(40):
    eng = HipNlp(st, model)
    eng.set_params(p)
    for cycle in range(3):
        out = (np.empty(1), np.empty((1, eng.n)), np.empty((1, eng.m)), np.empty((1, eng.nnz)))   # fresh arrays: freed below
        for _ in range(3):
            eng.eval(x, out=out)
            assert all(np.array_equal(a, b) for a, b in zip(out, want)), (rep, cycle)
        del out
        gc.collect()
        junk = [np.ones(eng.nnz) for _ in range(2)]   # the allocator hands the freed addresses out again
        del junk
    s = eng.host_stats()
    fallbacks += s["auto_fallbacks"]
    registered += s["auto_registered"]
    if rep % 2:
        eng.close()
    else:
        del eng          # destroyed by the garbage collector, after its arrays
    gc.collect()
print("ok: 40 handles x 3 array generations; registrations %d, verified fallbacks %d" % (registered, fallbacks))
