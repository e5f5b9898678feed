#!/usr/bin/env python3
"""GPU box: same-session A/B of the callback kernels with and without the constant entries of jac g (CCS handle: every entry staged and
stored; varying-first handle: VARY instantiations).  Interleaved repetitions, HIP events of the library, one JSON line per configuration.
    python tools/diag/vary_ab.py [reps]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hippopt_amd.hipnlp import HipNlp, build_info  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload, place_on_step_flanks  # noqa: E402

model = synthetic_ergocub()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
CONFIGS = [("periodic", 100, 1), ("periodic", 100, 4), ("periodic", 100, 64), ("periodic", 100, 1024), ("stairs", 200, 16), ("stairs", 100, 64)]
if os.environ.get("VARY_AB_CONFIGS"):
    CONFIGS = [(w, int(n), int(b)) for w, n, b in (c.split(":") for c in os.environ["VARY_AB_CONFIGS"].split(","))]
print(json.dumps({"build": build_info()}))
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
    engs = {"ccs": HipNlp(st, model, batch=B), "varying_first": HipNlp(st, model, batch=B, jac_varying_first=True)}
    bufs = {}
    with torch.cuda.stream(stream):
        xd = torch.from_numpy(x).to(dev)
        for k, e in engs.items():
            e.set_params(p)
            bufs[k] = [torch.empty(B * q, dtype=torch.float64, device=dev) for q in (1, e.n, e.m, e.nnz)]
    stream.synchronize()
    steps, ev = (40, 4) if N * B > 50000 else ((400, 16) if N * B > 400 else (1600, 16))
    res = {k: [] for k in engs}
    for rep in range(REPS):
        for k, e in engs.items():
            args = [t.data_ptr() for t in bufs[k]]
            for _ in range(steps // 4):
                e.eval_device(xd.data_ptr(), *args, stream=stream.cuda_stream)
            stream.synchronize()
            if e.kernels_per_eval() == 1 and N * B <= 400:
                e.profile_begin_runs(steps // (4 * ev), ev)
            else:
                e.profile_begin(steps // ev, ev)
            t0 = time.perf_counter()
            for _ in range(steps):
                e.eval_device(xd.data_ptr(), *args, stream=stream.cuda_stream)
            stream.synchronize()
            wall = (time.perf_counter() - t0) / steps
            kern_ms, _, _ = e.profile_end()
            res[k].append((1e3 * kern_ms, 1e6 * wall))
    line = {"workload": "%s N=%d x %d" % (wl, N, B), "knots": N * B}
    for k in engs:
        ku = sorted(r[0] for r in res[k])
        line[k] = {"kernel_us_median": ku[len(ku) // 2], "kernel_us_all": [round(v, 2) for v in ku], "wall_us_per_step_min": min(r[1] for r in res[k]),
                   "M_knots_per_s": N * B / ku[len(ku) // 2]}
    line["speedup"] = line["ccs"]["kernel_us_median"] / line["varying_first"]["kernel_us_median"]
    print(json.dumps(line), flush=True)
    for e in engs.values():
        e.close()
