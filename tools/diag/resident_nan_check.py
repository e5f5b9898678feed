"""GPU box diagnostic: the non-finite flag through the resident path, full and lifted layout, several calls in a row."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hippopt_amd.hipnlp import HipNlp
from hippopt_amd.kinodyn_settings import periodic_step_settings
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.synthetic import make_workload

model = synthetic_ergocub()
st = periodic_step_settings(100, model)
x, p = make_workload(st, model, batch=1, seed=4300)
bad = x.copy()
bad[0, 130:134] = 0.0
for lifted in (False, True):
    for resident in (False, True):
        eng = HipNlp(st, model, detect_simple_bounds=lifted)
        eng.set_params(p)
        eng.set_auto_register(False)
        if resident:
            eng.set_resident(300.0)
        rcs = []
        for i in range(6):
            xi = bad if i % 2 == 0 else x
            f, grad, g, jac = eng.eval(xi, nan_ok=True)
            rc = eng.lib.hipnlp_eval(eng.h, xi.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)), 1, None, None, None, None)
            rcs.append((int(rc), int(np.sum(~np.isfinite(g))), int(np.sum(~np.isfinite(jac))), bool(np.isfinite(f[0]))))
        print("lifted", lifted, "resident", resident, rcs, flush=True)
        eng.close()
