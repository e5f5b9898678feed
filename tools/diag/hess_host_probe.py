#!/usr/bin/env python3
"""GPU box: individual hipnlp_eval_hess calls through host buffers (the value array reused, as IPOPT's adapter does): per-call wall
clock and the handle's auto-registration counters, then the same with auto-registration off."""
import os
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
eng = HipNlp(st, model)
eng.set_params(p)
lam = np.random.RandomState(0).standard_normal((1, eng.m))
hv = eng.eval_hess(x, 1.0, lam)
for i in range(8):
    t = time.perf_counter()
    eng.eval_hess(x + 1e-4 * i, 1.0, lam, out=hv)
    print(i, "%.0f us" % ((time.perf_counter() - t) * 1e6), eng.host_stats(), flush=True)
eng.set_auto_register(False)
for i in range(4):
    t = time.perf_counter()
    eng.eval_hess(x + 1e-4 * i, 1.0, lam, out=hv)
    print("auto-registration off", i, "%.0f us" % ((time.perf_counter() - t) * 1e6), flush=True)

# the same sequence inside a process that has used torch and the device path first (bench.py's order)
if os.environ.get("HESS_WITH_TORCH", "1") == "1":
    import torch
    eng2 = HipNlp(st, model)
    eng2.set_params(p)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        xd = torch.tensor(x, device="cuda")
        ld = torch.tensor(lam, device="cuda")
        sd = torch.ones(1, dtype=torch.float64, device="cuda")
        od = torch.zeros((1, eng2.hess_nnz()), dtype=torch.float64, device="cuda")
        for _ in range(20):
            eng2.eval_hess_device(xd.data_ptr(), sd.data_ptr(), ld.data_ptr(), od.data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()
    hv2 = eng2.eval_hess(x, 1.0, lam)
    for i in range(6):
        t = time.perf_counter()
        eng2.eval_hess(x, 1.0, lam, out=hv2)
        print("after torch + device path", i, "%.0f us" % ((time.perf_counter() - t) * 1e6), eng2.host_stats(), flush=True)
