import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from hippopt_amd.hipnlp import HipPose
from hippopt_amd.pose_settings import make_pose_workload, pose_finder_settings
from hippopt_amd.robot_model import synthetic_ergocub
model=synthetic_ergocub(); st=pose_finder_settings(model)
x,p=make_pose_workload(st, model, 1, 3)
eng=HipPose(st, model, batch=1); eng.set_params(p)
lam=np.random.RandomState(0).standard_normal((1,eng.m)); sig=np.ones(1)
for name,fn in (("eval (f, grad, g, jac)", lambda: eng.eval(x)), ("eval_hess", lambda: eng.eval_hess(x, sig, lam))):
    for _ in range(200): fn()
    t0=time.perf_counter()
    for _ in range(2000): fn()
    print(name, "%.1f us per call" % ((time.perf_counter()-t0)/2000*1e6))
