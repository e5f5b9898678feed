#!/usr/bin/env python3
"""GPU box: where the time of the host-buffer path (hipnlp_eval) goes, per callback kind: raw ctypes calls (no numpy wrapper),
plain caller arrays against arrays registered with the library, and the library's own wall-clock breakdown of each call."""
import ctypes as C
import sys
import time
sys.path.insert(0, '/root/repo')
import numpy as np
from hippopt_amd.hipnlp import HipNlp
from hippopt_amd.kinodyn_settings import periodic_step_settings
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.synthetic import make_workload
md = synthetic_ergocub()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
st = periodic_step_settings(N, md)
x, p = make_workload(st, md, 1, 5)
eng = HipNlp(st, md)
eng.set_params(p)
eng.set_prefetch(())
rng = np.random.RandomState(1)
xs = [np.ascontiguousarray(x + 1e-3 * i * rng.standard_normal(x.shape)) for i in range(4)]
dp = C.POINTER(C.c_double)
outs = [np.zeros(1), np.zeros(eng.n), np.zeros(eng.m), np.zeros(eng.nnz)]
ptr = lambda a: a.ctypes.data_as(dp)  # noqa: E731
xp = [ptr(a) for a in xs]
kinds = {"f": (0,), "g": (2,), "grad": (1,), "jac": (3,), "f+g": (0, 2), "all": (0, 1, 2, 3)}


def run(label):
    for name, idx in kinds.items():
        args = [ptr(outs[i]) if i in idx else None for i in range(4)]
        for i in range(20):
            eng.lib.hipnlp_eval(eng.h, xp[i % 4], 1, *args)
        acc = np.zeros(4)
        reps = 300
        t0 = time.perf_counter()
        for i in range(reps):
            eng.lib.hipnlp_eval(eng.h, xp[i % 4], 1, *args)
        el = (time.perf_counter() - t0) / reps
        for i in range(50):
            eng.lib.hipnlp_eval(eng.h, xp[i % 4], 1, *args)
            acc += eng.host_breakdown()
        acc /= 50
        print("%-10s %-5s %7.1f us per call   [x staging %.1f | enqueue %.1f | wait %.1f | copy out %.1f]" % (label, name, 1e6 * el, *acc), flush=True)


run("plain")
eng.register_outputs(outs)
run("registered")
eng.unregister_outputs(outs)
eng.set_prefetch(("f", "grad", "g"))
a_f = [ptr(outs[0]), None, None, None]
a_g = [None, None, ptr(outs[2]), None]
a_gr = [None, ptr(outs[1]), None, None]
a_j = [None, None, None, ptr(outs[3])]
for reg in (False, True):
    if reg:
        eng.register_outputs(outs)
    t0 = time.perf_counter()
    for i in range(300):
        eng.lib.hipnlp_eval(eng.h, xp[i % 4], 1, *a_f)
        eng.lib.hipnlp_eval(eng.h, xp[i % 4], 0, *a_g)
        eng.lib.hipnlp_eval(eng.h, xp[i % 4], 0, *a_gr)
        eng.lib.hipnlp_eval(eng.h, xp[i % 4], 0, *a_j)
    print("ipopt iterate (f new x, then g, grad, jac cached), registered=%s: %.1f us" % (reg, 1e6 * (time.perf_counter() - t0) / 300))
eng.register_outputs(outs)
eng.set_early_outputs(True)
t0 = time.perf_counter()
for i in range(300):
    eng.lib.hipnlp_eval(eng.h, xp[i % 4], 1, *a_f)
    eng.lib.hipnlp_eval(eng.h, xp[i % 4], 0, *a_g)
    eng.lib.hipnlp_eval(eng.h, xp[i % 4], 0, *a_gr)
    eng.lib.hipnlp_eval(eng.h, xp[i % 4], 0, *a_j)
print("ipopt iterate, registered + early outputs (hipnlp_set_early_outputs): %.1f us" % (1e6 * (time.perf_counter() - t0) / 300))
eng.set_early_outputs(False)
eng.unregister_outputs(outs)
lam = np.random.RandomState(0).standard_normal((1, eng.m))
hv = eng.eval_hess(x, 1.0, lam)
t0 = time.perf_counter()
for _ in range(200):
    eng.eval_hess(x, 1.0, lam, out=hv)
print("eval_hess host path: %.1f us" % (1e6 * (time.perf_counter() - t0) / 200))
eng.register_outputs([hv])
for _ in range(20):
    eng.eval_hess(x, 1.0, lam, out=hv)
t0 = time.perf_counter()
for _ in range(200):
    eng.eval_hess(x, 1.0, lam, out=hv)
print("eval_hess host path, caller array registered: %.1f us" % (1e6 * (time.perf_counter() - t0) / 200))
ref = eng.eval_hess(x, 1.0, lam)
assert np.array_equal(ref, hv)
eng.unregister_outputs([hv])
