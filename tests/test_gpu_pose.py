"""GPU parity of the static pose finder (hipnlp_pose_*, BASELINE config 2) through the C-ABI against the CPU oracle and the
golden vectors of the reference's pose finder.  Tolerance: fp64, max |a-b| / max(1,|b|) <= 1e-11 (1e-10 against the fixtures)."""
import json
import os

import numpy as np
import pytest

from hippopt_amd.pose_settings import make_pose_workload
from test_golden_pose import GOLD, check_against_fixture, pose_settings_for
from test_pose_body_hostemu import variants

pytestmark = pytest.mark.gpu
TOL = 1e-11


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


@pytest.mark.parametrize("name", ["default", "constrained", "steps"])
def test_pose_matches_oracle(model, name):
    from hippopt_amd.hipnlp import HipPose
    from oracle_lib import PoseOracle
    st = variants(model)[name]
    B = 5
    x, p = make_pose_workload(st, model, B, 800)
    if name == "steps":
        rng = np.random.RandomState(2)
        for b in range(2):
            for c in range(8):
                x[b][6 * c] = 0.45 + rng.choice([-1.0, 1.0]) * 0.3 * rng.uniform(0.95, 1.01)
                x[b][6 * c + 1] = 0.3 * rng.uniform(-1, 1)
                x[b][6 * c + 2] = 0.1 + 0.05 * rng.standard_normal()
    eng, orc = HipPose(st, model, batch=B), PoseOracle(st, model)
    assert (eng.n, eng.m, eng.nnz, eng.np) == (orc.n, orc.m, orc.nnz, orc.np)
    assert eng.row_blocks() == orc.row_blocks()
    ir, jc = eng.sparsity()
    iro, jco = orc.sparsity()
    assert np.array_equal(ir, iro) and np.array_equal(jc, jco)
    eng.set_params(p)
    f, grad, g, jac = eng.eval(x)
    lb, ub = eng.bounds()
    names, terms = eng.cost_terms()
    assert names == orc.cost_term_names()
    for b in range(B):
        fo, grado, go, jaco = orc.eval(x[b], p[b])
        assert rel(f[b], fo) < TOL and rel(grad[b], grado) < TOL and rel(g[b], go) < TOL and rel(jac[b], jaco) < TOL
        assert np.allclose(terms[b], orc.cost_terms(), rtol=1e-12, atol=1e-12)
        lbo, ubo = orc.bounds(p[b])
        assert np.array_equal(lb[b], lbo) and np.array_equal(ub[b], ubo)
    f2, grad2, g2, jac2 = eng.eval(x)   # bitwise reproducible
    assert np.array_equal(f, f2) and np.array_equal(jac, jac2) and np.array_equal(g, g2) and np.array_equal(grad, grad2)


@pytest.mark.parametrize("name", ["pose_default", "pose_step_constrained"])
def test_pose_matches_reference_fixture(model, name):
    from hippopt_amd.hipnlp import HipPose
    z = np.load(os.path.join(GOLD, name + ".npz"))
    st = pose_settings_for(json.loads(str(z["meta"])), model)
    eng = HipPose(st, model)
    eng.set_params(z["p"][None, :])
    f, grad, g, jac = eng.eval(z["x"][None, :])
    lb, ub = eng.bounds()
    ir, jc = eng.sparsity()
    check_against_fixture(z, ir, jc, float(f[0]), grad[0], g[0], jac[0], lb[0], ub[0], tol=1e-10)


def test_pose_errors_and_large_batch(model):
    from hippopt_amd.hipnlp import HipNlpError, HipPose
    from oracle_lib import PoseOracle
    st = variants(model)["default"]
    eng = HipPose(st, model, batch=2)
    with pytest.raises(HipNlpError) as ei:
        eng.eval(np.zeros((2, 81)))
    assert ei.value.code == -4                      # parameters not set
    x, p = make_pose_workload(st, model, 2, 1)
    eng.set_params(p)
    x[1][51:55] = 0.0                               # zero quaternion -> normalisation divides by zero
    with pytest.raises(HipNlpError) as ei:
        eng.eval(x)
    assert ei.value.code == -5
    bad = variants(model)["default"]
    bad.terrain = 7
    with pytest.raises(HipNlpError):
        HipPose(bad, model)
    # a batch that fills the chip several times over; spot-check a few poses
    B = 4096
    xs, ps = make_pose_workload(st, model, 16, 3)
    xb, pb = np.tile(xs, (B // 16, 1)), np.tile(ps, (B // 16, 1))
    xb += 1e-3 * np.random.RandomState(0).standard_normal(xb.shape)
    big = HipPose(st, model, batch=B)
    big.set_params(pb)
    f, grad, g, jac = big.eval(xb)
    orc = PoseOracle(st, model)
    for b in (0, 17, 2049, B - 1):
        fo, grado, go, jaco = orc.eval(xb[b], pb[b])
        assert rel(f[b], fo) < TOL and rel(grad[b], grado) < TOL and rel(g[b], go) < TOL and rel(jac[b], jaco) < TOL
    print("pose kernel, batch %d: %.3f ms" % (B, big.last_kernel_ms()))
