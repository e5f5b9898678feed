#!/usr/bin/env python3
"""GPU box diagnostic: bench.py's per-kind host-visible loop (HipNlp.eval through the numpy wrapper) with the kinds in several
orders, with and without torch's HIP context initialised — to tell a property of a callback kind from a property of the loop."""
import sys
import time
sys.path.insert(0, '/root/repo')
import numpy as np
from hippopt_amd.hipnlp import HipNlp
from hippopt_amd.kinodyn_settings import periodic_step_settings
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.synthetic import make_workload
md = synthetic_ergocub()
st = periodic_step_settings(100, md)
x, p = make_workload(st, md, 1, 5)
eng = HipNlp(st, md)
eng.set_params(p)
eng.set_prefetch(())
rng = np.random.RandomState(1)
xs = [x + 1e-3 * i * rng.standard_normal(x.shape) for i in range(4)]
outs = eng.eval(x)
names = ("f", "grad", "g", "jac")


def loop(order, tag):
    for name in order:
        want = (name,)
        out = tuple(o if k in want else None for k, o in zip(names, outs))
        for i in range(10):
            eng.eval(xs[i % 4], want=want, out=out)
        acc = np.zeros(4)
        t0 = time.perf_counter()
        for i in range(200):
            eng.eval(xs[i % 4], want=want, out=out)
            acc += eng.host_breakdown()
        el = (time.perf_counter() - t0) / 200
        print("%-28s %-5s %7.1f us   library mean [%.1f %.1f %.1f %.1f]" % (tag, name, 1e6 * el, *(acc / 200)), flush=True)


loop(("f", "g", "grad", "jac"), "bench order")
loop(("grad", "jac", "f", "g"), "grad first")
outs2 = [np.zeros_like(o) for o in outs]
outs = outs2
loop(("f", "g", "grad", "jac"), "fresh zeros arrays")
import torch
torch.zeros(4, device="cuda")
torch.cuda.synchronize()
loop(("f", "g", "grad", "jac"), "after torch HIP init")
