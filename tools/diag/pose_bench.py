#!/usr/bin/env python3
"""Static pose finder (BASELINE config 2) on the GPU: device-resident throughput of the callback quartet (hipnlp_pose_eval_device)
and of the exact Hessian of the Lagrangian (hipnlp_pose_eval_hess_device) for a few batch sizes.  One JSON line per batch.
Under rocprofv3 --kernel-trace --stats this gives the kernel summaries committed as profiles/r01_pose_kernel_stats.csv."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hippopt_amd import hipnlp as _hipnlp
if os.environ.get("POSE_LIB"):   # another build of the library (diagnostic A/B)
    _hipnlp._LIB_PATH = os.path.join(ROOT, os.environ["POSE_LIB"])
from hippopt_amd.hipnlp import HipPose
from hippopt_amd.pose_settings import make_pose_workload, pose_finder_settings
from hippopt_amd.robot_model import synthetic_ergocub

model = synthetic_ergocub()
st = pose_finder_settings(model)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
WHAT = os.environ.get("POSE_WHAT", "")   # "callbacks" / "hessian": only that kernel (one configuration per rocprofv3 trace)
for B in [int(b) for b in os.environ.get("POSE_BATCHES", "1,256,4096").split(",")]:
    x, p = make_pose_workload(st, model, B, 11)
    eng = HipPose(st, model, batch=B)
    eng.set_params(p)
    hr, hc = eng.hess_sparsity()
    xd = torch.from_numpy(x).to(dev)
    f = torch.empty(B, dtype=torch.float64, device=dev)
    grad = torch.empty(B * eng.n, dtype=torch.float64, device=dev)
    g = torch.empty(B * eng.m, dtype=torch.float64, device=dev)
    jac = torch.empty(B * eng.nnz, dtype=torch.float64, device=dev)
    lam = torch.from_numpy(np.random.RandomState(1).standard_normal((B, eng.m))).to(dev)
    sig = torch.ones(B, dtype=torch.float64, device=dev)
    hess = torch.empty(B * hr.size, dtype=torch.float64, device=dev)
    steps = 2000 if B == 1 else (1000 if B == 256 else 200)
    res = {"workload": "pose finder, batch %d" % B, "n": eng.n, "m": eng.m, "nnz": eng.nnz, "nnz_h": int(hr.size)}
    for name, fn in (("callbacks", lambda: eng.eval_device(xd.data_ptr(), f.data_ptr(), grad.data_ptr(), g.data_ptr(), jac.data_ptr(), stream)),
                     ("hessian", lambda: eng.eval_hess_device(xd.data_ptr(), sig.data_ptr(), lam.data_ptr(), hess.data_ptr(), stream))):
        if WHAT and name != WHAT:
            continue
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res[name] = {"ms_per_launch": 1e3 * dt, "poses_per_s": B / dt}
    print(json.dumps(res))
