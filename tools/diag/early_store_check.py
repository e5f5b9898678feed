#!/usr/bin/env python3
"""GPU box: the host-path outputs with and without the early copy-out (HIPNLP_EARLY_STORE read by hipnlp_create), entry by entry."""
import os
# (the environment overrides below exist in the diagnostic build of the library only: __graft_entry__.build() -> tests/_build)
DIAG_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "_build", "libhipnlp_diag.so")
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hippopt_amd.hipnlp import HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload, place_on_step_flanks  # noqa: E402

model = synthetic_ergocub()
for name, maker in (("periodic", periodic_step_settings), ("stairs", stairs_settings)):
    for vf in (False, True):
        B = int(os.environ.get("CHECK_BATCH", "1"))
        st = maker(int(os.environ.get("CHECK_N", "24")), model)
        x, p = make_workload(st, model, batch=B, seed=91)
        engs = []
        for e in ("1", "0"):
            os.environ["HIPNLP_EARLY_STORE"] = e
            engs.append(HipNlp(st, model, batch=B, jac_varying_first=vf, library=DIAG_SO))
            engs[-1].set_params(p)
        del os.environ["HIPNLP_EARLY_STORE"]
        x = x + 1e-2 * np.random.RandomState(3).standard_normal(x.shape)
        a, b = engs[0].eval(x), engs[1].eval(x)
        ir, jc = engs[0].sparsity()
        for nm, u, v in zip(("f", "grad", "g", "jac"), a, b):
            bad = np.nonzero(u.ravel() != v.ravel())[0]
            print(name, "vf" if vf else "ccs", nm, "mismatches", bad.size, bad[:8], (u.ravel()[bad[:4]], v.ravel()[bad[:4]]) if bad.size else "")
            if nm == "jac" and bad.size:
                print("   rows", ir[bad[:12]], "cols", jc[bad[:12]], "cols mod 189", jc[bad[:12]] % 189)
            if nm == "g" and bad.size:
                print("   rows", bad[:12], engs[0].row_blocks()[:0])
