#!/usr/bin/env python3
"""GPU box: the four-wave VARY launches with the first workgroups of the grid staggered at entry (HIPNLP_STAGGER=first,units on the
diagnostic build: workgroups [0, first) wait slot x units x 2048 cycles, slot = the wave's slot number on its SIMD) against the plain launch.
Interleaved repetitions, the library's HIP events.   STAGGER_CONFIGS=periodic:100:64,...   STAGGER_SETTINGS=0:0,1280:1,..."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from diag_lib import diag_overrides  # noqa: E402
from hippopt_amd.hipnlp import HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload, place_on_step_flanks  # noqa: E402

model = synthetic_ergocub()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
CONFIGS = [(w, int(n), int(b)) for w, n, b in (c.split(":") for c in os.environ.get("STAGGER_CONFIGS", "periodic:100:64,stairs:200:16,periodic:100:16,periodic:100:256,periodic:100:1024").split(","))]
SETTINGS = [tuple(int(v) for v in s.split(":")) for s in os.environ.get("STAGGER_SETTINGS", "0:0,1280:1,1280:2,1280:3,640:2,1280:4").split(",")]
for wl, N, B in CONFIGS:
    st = (stairs_settings if wl == "stairs" else periodic_step_settings)(N, model)
    x1, p1 = make_workload(st, model, batch=1, seed=1004)
    if wl == "stairs":
        place_on_step_flanks(x1, st, seed=1004)
    x = x1 + 0.02 * np.random.RandomState(1005).standard_normal((B, x1.shape[1]))
    if wl == "stairs":
        cols = (189 * np.arange(N)[:, None] + np.array([15 * c + 6 + i for c in range(8) for i in range(3)] + [180, 181])[None, :]).reshape(-1)
        x[:, cols] = x1[0, cols][None, :] + 1e-3 * np.random.RandomState(1006).standard_normal((B, cols.size))
    p = np.tile(p1, (B, 1))
    engs = {}
    for first, units in SETTINGS:
        with diag_overrides(HIPNLP_STAGGER="%d,%d" % (first, units)) as lib:
            engs["%d:%d" % (first, units)] = HipNlp(st, model, batch=B, jac_varying_first=True, library=lib)
    bufs = {}
    with torch.cuda.stream(stream):
        xd = torch.from_numpy(x).to(dev)
        for k, e in engs.items():
            e.set_params(p)
            bufs[k] = [torch.zeros(B * q, dtype=torch.float64, device=dev) for q in (1, e.n, e.m, e.nnz)]
    stream.synchronize()
    steps, ev = (40, 4) if N * B > 50000 else (400, 16)
    res = {k: [] for k in engs}
    for rep in range(REPS):
        for k, e in engs.items():
            args = [t.data_ptr() for t in bufs[k]]
            for _ in range(steps // 4):
                e.eval_device(xd.data_ptr(), *args, stream=stream.cuda_stream)
            stream.synchronize()
            e.profile_begin(steps // ev, ev)
            for _ in range(steps):
                e.eval_device(xd.data_ptr(), *args, stream=stream.cuda_stream)
            stream.synchronize()
            kern_ms, _, _ = e.profile_end()
            res[k].append(1e3 * kern_ms)
    ref = [t.cpu().numpy() for t in bufs["0:0"]]
    line = {"workload": "%s N=%d x %d" % (wl, N, B)}
    for k in engs:
        ku = sorted(res[k])
        same = all(np.array_equal(a, t.cpu().numpy()) for a, t in zip(ref, bufs[k]))
        line[k] = [round(ku[len(ku) // 2], 2), round(N * B / ku[len(ku) // 2], 1), bool(same)]
    print(json.dumps(line), flush=True)
    for e in engs.values():
        e.close()
