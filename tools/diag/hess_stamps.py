#!/usr/bin/env python3
"""Diagnostic (never part of the product): per-wave timeline of the exact-Hessian kernel from s_memtime stamps of a -DHIPNLP_STAMPS
build (tools/diag/_build/libhipnlp_stamps.so, built by `tools/diag/stamps.py build`): end of every task group in program order,
arrival at / departure from every barrier.     usage (GPU box): hess_stamps.py [batch] [stairs]"""
import ctypes as C
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SO = os.environ.get("STAMPS_SO") or os.path.join(ROOT, "tools", "diag", "_build", "libhipnlp_stamps.so")


def program_tasks():
    """task names per wave and phase, parsed from the program table of knot_hess_body.h"""
    src = open(os.path.join(ROOT, "hippopt_amd", "csrc", "knot_hess_body.h")).read()
    phases = []
    for name in ("1A", "1B", "1C", "1D", "2", "3"):
        body = src[src.index("#define HIPNLP_KNOT_HESS_PHASE%s(" % name):]
        body = body[:body.index("BARRIER\n")]
        phases.append([(int(w), fn) for w, fn in re.findall(r"(?:KIN|RH)\((\d), ([\w<>]+),", body)])
    return phases


if __name__ == "__main__":
    from hippopt_amd import hipnlp
    hipnlp._LIB_PATH = SO
    import torch
    from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings
    from hippopt_amd.robot_model import synthetic_ergocub
    from hippopt_amd.synthetic import make_workload
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    stairs = len(sys.argv) > 2 and sys.argv[2] == "stairs"
    N = 100
    model = synthetic_ergocub()
    st = stairs_settings(N, model) if stairs else periodic_step_settings(N, model)
    x, p = make_workload(st, model, batch, 1004)
    eng = hipnlp.HipNlp(st, model, batch=batch)
    eng.set_params(p)
    hn = eng.hess_nnz()
    torch.cuda.set_stream(torch.cuda.Stream())
    xd = torch.tensor(x, device="cuda")
    ld = torch.tensor(np.random.RandomState(0).standard_normal((batch, eng.m)), device="cuda")
    sd = torch.ones(batch, dtype=torch.float64, device="cuda")
    out = torch.zeros((batch, hn), dtype=torch.float64, device="cuda")
    for _ in range(20):
        eng.eval_hess_device(xd.data_ptr(), sd.data_ptr(), ld.data_ptr(), out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    raw = np.zeros((2 * N * batch, 8, 128), np.uint64)[: 2 * N * batch]   # (hipnlp_debug_stamps copies the room of a SPLIT launch: two workgroups per knot)
    eng.lib.hipnlp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    eng.lib.hipnlp_debug_stamps(eng.h, raw.ctypes.data_as(C.c_void_p))
    o = raw[: N * batch].astype(np.int64)[:, :4]
    t0 = o[:, :, 0].min(axis=1, keepdims=True)                       # entry of the workgroup's first wave
    nb = int(o[0, 0, 2])
    total = (o[:, :, 4].max(axis=1) - t0[:, 0])
    print("exact Hessian kernel, %s, N=%d x batch %d: %d workgroups, %d barriers; workgroup entry -> last wave done, s_memtime cycles (shader clock, ~2.4 GHz):"
          % ("stairs" if stairs else "periodic", N, batch, len(o), nb))
    print("   median %.0f   p10 %.0f   p90 %.0f   max %.0f" % tuple(np.percentile(total, q) for q in (50, 10, 90, 100)))
    phases = program_tasks()
    med = lambda a: float(np.median(a))   # noqa: E731
    print("   staging done (median over workgroups, slowest wave): %.0f cycles" % med((o[:, :, 1] - t0).max(axis=1)))
    prev = o[:, :, 1]
    for ph in range(nb):
        arr, dep = o[:, :, 8 + 2 * ph], o[:, :, 9 + 2 * ph]
        print("   phase %-2s: slowest wave arrives %.0f cycles after the phase began; per wave busy [%s]" %
              (("1A", "1B", "1C", "1D", "2", "3")[ph], med((arr - prev).max(axis=1)), ", ".join("%.0f" % med(arr[:, w] - prev[:, w]) for w in range(4))))
        prev = dep
    print("   copy-out (last barrier -> end, slowest wave): %.0f cycles" % med((o[:, :, 4] - prev).max(axis=1)))
    # per task group: duration = its end stamp - previous stamp of the same wave (task end or barrier departure)
    print("   task groups (median cycles):")
    idx = [0, 0, 0, 0]
    prev = o[:, :, 1].copy()
    for ph in range(nb):
        for w, fn in phases[ph]:
            end = o[:, w, 32 + idx[w]]
            print("      phase %-2s wave %d  %-28s %.0f" % (("1A", "1B", "1C", "1D", "2", "3")[ph], w, fn, med(end - prev[:, w])))
            prev[:, w] = end
            idx[w] += 1
        prev = o[:, :, 9 + 2 * ph].copy()
