#!/usr/bin/env python3
"""GPU box: time hipnlp_eval_hess_device (exact Hessian of the Lagrangian) beside the callback quartet.  One JSON line per batch."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hippopt_amd.hipnlp import HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402

N = int(os.environ.get("HESS_N", "100"))
model = synthetic_ergocub()
torch_stream = torch.cuda.Stream()   # (a non-default stream: the library maps a null stream pointer to its own stream)
torch.cuda.set_stream(torch_stream)
for B in [int(b) for b in os.environ.get("HESS_BATCHES", "1,16,64,256").split(",")]:
    st = stairs_settings(N, model) if os.environ.get("HESS_WORKLOAD", "periodic") == "stairs" else periodic_step_settings(N, model)
    x, p = make_workload(st, model, batch=B, seed=5)
    eng = HipNlp(st, model, batch=B)
    eng.set_params(p)
    hn = eng.hess_nnz()
    xd = torch.tensor(x, device="cuda")
    ld = torch.tensor(np.random.RandomState(0).standard_normal((B, eng.m)), device="cuda")
    sd = torch.ones(B, dtype=torch.float64, device="cuda")
    out = torch.zeros((B, hn), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    steps = 200 if B <= 64 else 40
    for _ in range(10):
        eng.eval_hess_device(xd.data_ptr(), sd.data_ptr(), ld.data_ptr(), out.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.eval_hess_device(xd.data_ptr(), sd.data_ptr(), ld.data_ptr(), out.data_ptr(), stream=stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    bytes_per_knot = 8 * (189 + 79 + 274 + hn / N)   # x, p, lambda in; triplet values out
    print(json.dumps({"workload": "exact Hessian of the Lagrangian, %s, N=%d x batch %d" % (os.environ.get("HESS_WORKLOAD", "periodic"), N, B), "ms_per_eval": ms, "knots_per_s": N * B / (ms * 1e-3),
                      "nnz_h": hn, "algorithmic_bytes_per_knot": bytes_per_knot, "GBps": bytes_per_knot * N * B / (ms * 1e-3) / 1e9,
                      "note": "device pointers, no PCIe; one kernel launch per evaluation"}), flush=True)
