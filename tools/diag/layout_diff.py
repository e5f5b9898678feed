#!/usr/bin/env python3
"""GPU box diagnostic: the four-wave (compact LDS layout) kernel against the eight-wave (full layout) kernel and the oracle on the
stairs configuration: where and by how much do they differ."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from hippopt_amd.hipnlp import HipNlp
from hippopt_amd.kinodyn_settings import stairs_settings
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.synthetic import make_workload, place_on_step_flanks
from oracle_lib import Oracle
m = synthetic_ergocub()
N, B = 200, 16
st = stairs_settings(N, m)
bx, bp = make_workload(st, m, batch=1, seed=1005)
x = np.repeat(bx, B, axis=0)
for b in range(B):
    x[b] += 0.02 * np.random.RandomState(2000 + b).standard_normal(x.shape[1])
place_on_step_flanks(x[:2], st, seed=5)
p = np.repeat(bp, B, axis=0)
eng = HipNlp(st, m, batch=B); eng.set_params(p)
f, grad, g, jac = eng.eval(x)
one = HipNlp(st, m, batch=1); one.set_params(p[:1])
orc = Oracle(st, m)
ir, jc = eng.sparsity()
for b in range(B):
    f1, grad1, g1, jac1 = one.eval(x[b:b + 1])
    d = np.abs(jac1[0] - jac[b])
    bad = np.nonzero(d > 0)[0]
    fo, grado, go, jaco = orc.eval(x[b], p[b])
    rel = lambda a, r: float(np.max(np.abs(a - r) / np.maximum(1.0, np.abs(r))))  # noqa: E731
    print("b=%2d  jac differing entries %5d  max |diff| %.3e  max rel to value %.3e | vs oracle: 4-wave %.2e  8-wave %.2e | g equal %s grad equal %s" % (
        b, bad.size, d.max(), float(np.max(d / np.maximum(1e-300, np.abs(jac1[0])))) if bad.size else 0.0, rel(jac[b], jaco), rel(jac1[0], jaco),
        np.array_equal(g1[0], g[b]), np.array_equal(grad1[0], grad[b])), flush=True)
    if bad.size and b < 4:
        rows = ir[bad]; cols = jc[bad] % 189
        import collections
        print("     columns (within knot) of differing entries:", collections.Counter(cols.tolist()).most_common(12))
        print("     knots:", sorted(set((jc[bad] // 189).tolist()))[:20])
