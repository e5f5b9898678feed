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
                           "-o", SO, os.path.join(ROOT, "hippopt_amd", "csrc", "hipnlp.hip"), os.path.join(ROOT, "hippopt_amd", "csrc", "hipnlp_pose.hip")])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        sys.exit(0)
    from hippopt_amd import hipnlp
    hipnlp._LIB_PATH = SO
    from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings
    from hippopt_amd.robot_model import synthetic_ergocub
    from hippopt_amd.synthetic import make_workload
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    model = synthetic_ergocub()
    stairs = len(sys.argv) > 2 and sys.argv[2] == "stairs"
    st = stairs_settings(100, model) if stairs else periodic_step_settings(100, model)
    x, p = make_workload(st, model, batch, 1004)
    if stairs:
        from hippopt_amd.synthetic import place_on_step_flanks
        place_on_step_flanks(x, st, seed=1)
    eng = hipnlp.HipNlp(st, model, batch=batch)
    eng.set_params(p)
    for _ in range(20):
        eng.eval(x)
    import re
    prog = open(os.path.join(ROOT, "hippopt_amd", "csrc", "knot_body.h")).read()
    prog = prog[prog.index("#define HIPNLP_KNOT_PROGRAM"):]
    groups = re.findall(r"R\((\d), (\d), (\w+),", prog)
    out = np.zeros((100 * batch, 8, 64, 2), np.uint64)
    eng.lib.hipnlp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    eng.lib.hipnlp_debug_stamps(eng.h, out.ctypes.data_as(C.c_void_p))
    blk = out[len(out) // 2]  # an interior knot
    waves = [w for w in range(8) if int(blk[w, 0, 0]) == 999]
    t0 = min(int(blk[w, 0, 1]) for w in waves)
    print("interior knot: per wave, end time of each group / arrival at each barrier (cycles since block start) and duration")
    for w in waves:
        n = int(out[len(out) // 2, w, 63, 0])
        prev = int(blk[w, 0, 1]) - t0
        line = []
        for i in range(1, n):
            gid, tm = int(blk[w, i, 0]), int(blk[w, i, 1]) - t0
            name = groups[gid][2] if gid < 1000 else ("|B%d" % (gid - 1000) if gid < 2000 else "END")
            line.append("%s %d(+%d)" % (name, tm, tm - prev))
            prev = tm
        print("wave %d: " % w + "  ".join(line))
    print("kernel ms:", eng.last_kernel_ms())
