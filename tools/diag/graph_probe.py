#!/usr/bin/env python3
"""GPU box: are K back-to-back callback launches cheaper as ONE hipGraph launch?  Captures K hipnlp_eval_device calls (torch's stream
capture around the library's launches) and times replays against the same K launches issued on the stream."""
import json
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

K = int(os.environ.get("GRAPH_K", "20"))
model = synthetic_ergocub()
st = periodic_step_settings(100, model)
x, p = make_workload(st, model, batch=1, seed=3)
eng = HipNlp(st, model)
eng.set_params(p)
dev = torch.device("cuda", 0)
xs = [torch.tensor(x[0] + 1e-3 * i, device=dev) for i in range(4)]
f = torch.zeros(1, dtype=torch.float64, device=dev)
grad = torch.zeros(eng.n, dtype=torch.float64, device=dev)
g = torch.zeros(eng.m, dtype=torch.float64, device=dev)
jac = torch.zeros(eng.nnz, dtype=torch.float64, device=dev)
s = torch.cuda.Stream()


def launches(stream):
    for i in range(K):
        eng.eval_device(xs[i % 4].data_ptr(), f.data_ptr(), grad.data_ptr(), g.data_ptr(), jac.data_ptr(), stream=stream.cuda_stream)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    s.synchronize()
    return e0.elapsed_time(e1) / reps / K * 1e3


with torch.cuda.stream(s):
    launches(s)
s.synchronize()
ref = [t.clone() for t in (f, grad, g, jac)]
plain = timed(lambda: launches(s))
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s):
    launches(s)
rep = timed(graph.replay)
same = all(bool(torch.equal(a, b)) for a, b in zip(ref, (f, grad, g, jac)))
print(json.dumps({"K": K, "us_per_step_stream_launches": plain, "us_per_step_graph_replay": rep, "outputs_of_the_last_step_equal": same}))
