#!/usr/bin/env python3
"""GPU box: wall-clock per call of the host-buffer paths (hipnlp_eval, hipnlp_eval_hess) through the Python binding, with fresh
and with reused output arrays."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from hippopt_amd.hipnlp import HipNlp
from hippopt_amd.kinodyn_settings import periodic_step_settings
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.synthetic import make_workload
md = synthetic_ergocub()
st = periodic_step_settings(100, md)
x, p = make_workload(st, md, 1, 5)
eng = HipNlp(st, md); eng.set_params(p)
lam = np.random.RandomState(0).standard_normal((1, eng.m))
for _ in range(5): eng.eval_hess(x, 1.0, lam); eng.eval(x)
t0 = time.perf_counter()
for _ in range(200): eng.eval_hess(x, 1.0, lam)
t1 = time.perf_counter()
for _ in range(200): eng.eval(x)
t2 = time.perf_counter()
outs = eng.eval(x); hv = eng.eval_hess(x, 1.0, lam)
t3 = time.perf_counter()
for _ in range(200): eng.eval_hess(x, 1.0, lam, out=hv)
t4 = time.perf_counter()
for _ in range(200): eng.eval(x, out=outs)
t5 = time.perf_counter()
print("caller-owned arrays reused: eval_hess %.1f us, eval %.1f us" % (1e6 * (t4 - t3) / 200, 1e6 * (t5 - t4) / 200))
print("eval_hess host path %.1f us per call; eval host path %.1f us per call" % (1e6 * (t1 - t0) / 200, 1e6 * (t2 - t1) / 200))
fo = eng.eval(x, want=("f",))
t6 = time.perf_counter()
for _ in range(200): eng.eval(x, want=("f",), out=(fo[0], None, None, None))
t7 = time.perf_counter()
print("eval, only f copied to the caller (evaluation + the whole device-to-host block): %.1f us" % (1e6 * (t7 - t6) / 200))
import torch
xd = torch.tensor(x, device="cuda")
torch.cuda.synchronize()
t8 = time.perf_counter()
for _ in range(200):
    eng.eval_device(xd.data_ptr())
    torch.cuda.synchronize()
t9 = time.perf_counter()
print("eval_device + synchronize (no copies): %.1f us" % (1e6 * (t9 - t8) / 200))
