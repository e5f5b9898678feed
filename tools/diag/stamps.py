#!/usr/bin/env python3
"""Diagnostic (never part of the product): per-phase cycle shares of the knot kernel from s_memtime stamps
taken at every workgroup barrier by a -DHIPNLP_STAMPS build (tools/diag/_build/libhipnlp_stamps.so)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "diag", "_build", "libhipnlp_stamps.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DHIPNLP_STAMPS",
                           "-o", SO, os.path.join(ROOT, "hippopt_amd", "csrc", "hipnlp.hip")])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        sys.exit(0)
    from hippopt_amd import hipnlp
    hipnlp._LIB_PATH = SO
    from hippopt_amd.kinodyn_settings import periodic_step_settings
    from hippopt_amd.robot_model import synthetic_ergocub
    from hippopt_amd.synthetic import make_workload
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    model = synthetic_ergocub()
    st = periodic_step_settings(100, model)
    x, p = make_workload(st, model, batch, 1004)
    eng = hipnlp.HipNlp(st, model, batch=batch)
    eng.set_params(p)
    for _ in range(20):
        eng.eval(x)
    out = np.zeros((100 * batch, 4, 16), np.uint64)
    eng.lib.hipnlp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    eng.lib.hipnlp_debug_stamps(eng.h, out.ctypes.data_as(C.c_void_p))
    n = int(out[0, 0, 15])
    t = out[:, :, :n + 1].astype(np.int64)
    t0 = t[:, :, 0].min(axis=1)[:, None, None]
    rel = t - t0                      # [block][wave][stamp]: arrival of each wave at barrier s (stamp 0 = after the load barrier)
    names = ["start", "A", "B fk|hdyn|foot", "C links|frames|ends", "D composite|pkin", "F columns", "G kinc|comc|cmmc|feetd", "end (copy-out)"]
    med = np.median(rel, axis=0)      # [wave][stamp]
    print("arrival (cycles since block start) of wave 0..3 at each barrier; the latest wave bounds the phase")
    prev = 0.0
    for sidx in range(n + 1):
        row = med[:, sidx]
        print("%-26s w0 %7.0f  w1 %7.0f  w2 %7.0f  w3 %7.0f   phase %6.0f" % (names[sidx] if sidx < len(names) else "s%d" % sidx, row[0], row[1], row[2], row[3], row.max() - prev))
        prev = row.max()
    print("kernel ms:", eng.last_kernel_ms())
