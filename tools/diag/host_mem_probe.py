#!/usr/bin/env python3
"""GPU box: where the host-buffer callback's kernel time goes — the 100-knot launch (varying-first handle) with x and / or the outputs in
pinned host memory instead of HBM, HIP events around 200 launches each."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hippopt_amd.hipnlp import HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402

model = synthetic_ergocub()
st = periodic_step_settings(100, model)
x, p = make_workload(st, model, batch=1, seed=1)
eng = HipNlp(st, model, jac_varying_first=True)
eng.set_params(p)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream()


def buf(n, host):
    return torch.zeros(n, dtype=torch.float64).pin_memory() if host else torch.zeros(n, dtype=torch.float64, device=dev)


for xh in (False, True):
    for which in ("none", "g", "grad", "jac", "all"):
        xd = buf(eng.n, xh)
        xd.copy_(torch.from_numpy(x[0]))
        outs = {k: buf(sz, which in (k, "all")) for k, sz in (("f", 1), ("grad", eng.n), ("g", eng.m), ("jac", eng.nnz))}
        args = (xd.data_ptr(), outs["f"].data_ptr(), outs["grad"].data_ptr(), outs["g"].data_ptr(), outs["jac"].data_ptr())
        with torch.cuda.stream(stream):
            for _ in range(20):
                eng.eval_device(*args, stream=stream.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(200):
                eng.eval_device(*args, stream=stream.cuda_stream)
            e1.record(stream)
        stream.synchronize()
        print("x in %s, host outputs: %-5s  %.2f us per launch" % ("pinned host memory" if xh else "HBM", which, 1e3 * e0.elapsed_time(e1) / 200), flush=True)
