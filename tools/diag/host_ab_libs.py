#!/usr/bin/env python3
"""GPU box: the host-buffer paths of two (or more) BUILDS of the library in one process, alternating pass by pass — hipnlp_eval with all four
outputs into reused caller arrays (varying-first handle: what HipNlpSolver and the IPOPT binding create), f alone, and hipnlp_eval_hess into a
reused value array.  Outputs of the builds compared bit for bit.
    HOST_AB_LIBS=name=path,name2=path2 python tools/diag/host_ab_libs.py        (HOST_AB_WORKLOAD=periodic|stairs, HOST_AB_N=100)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hippopt_amd.hipnlp import HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload, place_on_step_flanks  # noqa: E402

N = int(os.environ.get("HOST_AB_N", "100"))
stairs = os.environ.get("HOST_AB_WORKLOAD") == "stairs"
model = synthetic_ergocub()
st = (stairs_settings if stairs else periodic_step_settings)(N, model)
x, p = make_workload(st, model, batch=1, seed=1)
if stairs:
    place_on_step_flanks(x, st, seed=1)
xs = [x + 1e-3 * i for i in range(4)]
libs = [spec.split("=") for spec in os.environ["HOST_AB_LIBS"].split(",")]
engs = {name: HipNlp(st, model, jac_varying_first=True, library=path) for name, path in libs}
for e in engs.values():
    e.set_params(p)
    e.set_prefetch(())
lam = np.random.RandomState(0).standard_normal((1, next(iter(engs.values())).m))
outs = {n: e.eval(x) for n, e in engs.items()}
hess = {n: e.eval_hess(x, 1.0, lam).copy() for n, e in engs.items()}
names = list(engs)
for n in names[1:]:
    assert all(np.array_equal(a, b) for a, b in zip(outs[names[0]], outs[n])), n
    assert np.array_equal(hess[names[0]], hess[n]), n


def passes(fn, reps=5, calls=200):
    best = {n: 1e9 for n in engs}
    for n in engs:
        for i in range(20):
            fn(n, i)
    for _ in range(reps):
        for n in engs:
            t0 = time.perf_counter()
            for i in range(calls):
                fn(n, i)
            best[n] = min(best[n], (time.perf_counter() - t0) / calls)
    return best


legs = {
    "all four outputs": lambda n, i: engs[n].eval(xs[i % 4], out=outs[n]),
    "f alone": lambda n, i: engs[n].eval(xs[i % 4], want=("f",)),
    "exact Hessian": lambda n, i: engs[n].eval_hess(xs[i % 4], 1.0, lam, out=hess[n]),
}
print("%s N = %d, us per call (best of 5 passes of 200 calls, builds alternating)" % ("smooth steps" if stairs else "planar", N))
for leg, fn in legs.items():
    b = passes(fn)
    print("  %-18s " % leg + "   ".join("%s %.2f" % (n, 1e6 * b[n]) for n in engs), flush=True)
for n in names[1:]:
    assert all(np.array_equal(a, b) for a, b in zip(outs[names[0]], outs[n])), n
    assert np.array_equal(hess[names[0]], hess[n]), n
print("  outputs of the builds: bit for bit the same")
