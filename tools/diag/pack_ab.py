#!/usr/bin/env python3
"""GPU box: the two-knots-per-workgroup lane-packed callback kernel (HIPNLP_PACK=1, hipnlp_knot_kernel_x2) against the four-wave
kernel: outputs bit for bit (same arithmetic, same summation tree) and launch time, one JSON line per configuration."""
import json
import os
# (the environment overrides below exist in the diagnostic build of the library only: __graft_entry__.build() -> tests/_build)
DIAG_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "_build", "libhipnlp_diag.so")
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hippopt_amd.hipnlp import G_STAGE, HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings, stairs_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402

model = synthetic_ergocub()
torch.cuda.set_stream(torch.cuda.Stream())
CONFIGS = [("periodic", 100, 64, {}), ("periodic", 100, 1024, {}), ("stairs", 200, 16, {}), ("periodic", 31, 40, {}),
           ("single", 30, 24, {}), ("periodic", 100, 8, {"knot_begin": 13, "knot_end": 58}), ("periodic", 300, 4, {})]
MAKERS = {"periodic": periodic_step_settings, "single": single_step_settings, "stairs": stairs_settings}


def engine(st, B, pack, kw):
    os.environ["HIPNLP_PACK"] = str(pack)
    os.environ["HIPNLP_WAVES"] = "4"
    try:
        return HipNlp(st, model, batch=B, library=DIAG_SO, **kw)
    finally:
        del os.environ["HIPNLP_PACK"], os.environ["HIPNLP_WAVES"]


for name, N, B, kw in CONFIGS:
    st = MAKERS[name](N, model)
    x, p = make_workload(st, model, batch=B, seed=77)
    xd = torch.tensor(x, device="cuda")
    res = {}
    outs = {}
    for pack in (0, 4, 8):
        eng = engine(st, B, pack, kw)
        eng.set_params(p)
        nk = kw.get("knot_end", N) - kw.get("knot_begin", 0)
        shard = bool(kw)
        f = torch.full((B,), -1.0, dtype=torch.float64, device="cuda")
        gsz = (B * nk * G_STAGE) if shard else B * eng.m
        jsz = (B * int(eng.dims.shard_nnz)) if shard else B * eng.nnz
        grsz = (B * int(eng.dims.shard_grad)) if shard else B * eng.n
        grad = torch.full((grsz,), -1.0, dtype=torch.float64, device="cuda")
        g = torch.full((gsz,), -1.0, dtype=torch.float64, device="cuda")
        jac = torch.full((jsz,), -1.0, dtype=torch.float64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream

        def step():
            if shard:
                eng.eval_device_shard(xd.data_ptr(), f.data_ptr(), grad.data_ptr(), g.data_ptr(), jac.data_ptr(), stream=stream)
            else:
                eng.eval_device(xd.data_ptr(), f.data_ptr(), grad.data_ptr(), g.data_ptr(), jac.data_ptr(), stream=stream)
        steps = 200 if N * B <= 8000 else 40
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        res[pack] = e0.elapsed_time(e1) / steps * 1e3
        outs[pack] = [t.cpu().numpy() for t in (f, grad, g, jac)]
        eng.close()
    same = [bool(np.array_equal(a, b, equal_nan=True) and np.array_equal(a, c, equal_nan=True)) for a, b, c in zip(outs[0], outs[4], outs[8])]
    for nm, a, b in [(n_ + str(v), a_, outs[v][i]) for v in (4, 8) for i, (n_, a_) in enumerate(zip(("f", "grad", "g", "jac"), outs[0]))]:
        bad = np.nonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))[0]
        if bad.size:
            per = a.size // B
            print("MISMATCH", nm, bad.size, "of", a.size, "first", [(int(i // per), int(i % per), float(a[i]), float(b[i])) for i in bad[:12]], flush=True)
    print(json.dumps({"config": "%s N=%d x %d %s" % (name, N, B, kw or ""), "us_four_wave": res[0], "us_packed4": res[4], "us_packed8": res[8], "speedup_packed4": res[0] / res[4], "speedup_packed8": res[0] / res[8],
                      "Mknots_s": {v: (kw.get("knot_end", N) - kw.get("knot_begin", 0)) * B / res[v] for v in res},
                      "bitwise_f_grad_g_jac": same}), flush=True)
