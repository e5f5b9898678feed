import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hippopt_amd.hipnlp import HipNlp
from hippopt_amd.kinodyn_settings import single_step_settings
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.synthetic import make_workload
model = synthetic_ergocub()
st = single_step_settings(30, model)
x, p = make_workload(st, model, batch=1, seed=3)
for lifted in (False, True):
    eng = HipNlp(st, model, detect_simple_bounds=lifted)
    eng.set_params(p)
    lam = np.random.RandomState(0).standard_normal((1, eng.m))
    ts = []
    for i in range(12):
        t = time.perf_counter(); eng.eval_hess(x + 1e-3 * i, 1.0, lam); ts.append((time.perf_counter() - t) * 1e6)
    print("lifted", lifted, ["%.0f" % t for t in ts])
