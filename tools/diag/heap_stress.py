#!/usr/bin/env python3
"""GPU box: the pattern of tests/test_gpu_parity.py::test_early_outputs_into_registered_arrays where the full suite once died inside malloc
(numpy empty_like right behind two HipNlp.eval calls on fresh arrays): small horizons (outputs of 10 - 200 KB: heap-arena memory, not
mmap), library-allocated outputs whose addresses the allocator hands out again, fresh handles, allocations in between.
HEAP_STRESS_REPS handles (default 300)."""
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
reps = int(os.environ.get("HEAP_STRESS_REPS", "300"))
regs = falls = 0
for rep in range(reps):
    N = (9, 12, 7, 16, 24)[rep % 5]
    st = periodic_step_settings(N, model)
    x, p = make_workload(st, model, batch=1, seed=43 + rep)
    x2 = x + 1e-2
    eng = HipNlp(st, model, jac_varying_first=bool(rep & 1))
    eng.set_params(p)
    ref = {0: [a.copy() for a in eng.eval(x)], 1: [a.copy() for a in eng.eval(x2)]}
    outs = [np.zeros_like(a) for a in ref[0]]
    for i in range(3):
        got = eng.eval(x if i % 2 == 0 else x2)
        assert all(np.array_equal(u, v) for u, v in zip(got, ref[i % 2])), (rep, i)
        junk = [np.zeros(n) for n in (1, 17, eng.n, eng.m, eng.nnz)]
        del got, junk
    s = eng.host_stats()
    regs += s["auto_registered"]
    falls += s["auto_fallbacks"]
    if rep % 3 == 0:
        eng.close()
    del eng, ref, outs
    if rep % 7 == 0:
        gc.collect()
    if rep % 50 == 0:
        print("rep", rep, "registrations", regs, "fallbacks", falls, flush=True)
print("ok: %d handles; registrations %d, verified fallbacks %d" % (reps, regs, falls))
