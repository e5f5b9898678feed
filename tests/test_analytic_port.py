"""The CPU baseline of bench.py (oracle/analytic_port.cpp: the engine's analytic knot program on the host) must
compute what the oracle computes — a baseline that times something else would be worthless — and must not depend on the number of
threads it is run with."""
import numpy as np
import pytest

from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings
from hippopt_amd.synthetic import make_workload, place_on_step_flanks


@pytest.mark.parametrize("maker,horizon", [(periodic_step_settings, 12), (stairs_settings, 6)])
def test_analytic_port_matches_oracle(model, maker, horizon):
    from analytic_port_lib import AnalyticPort
    from oracle_lib import Oracle
    st = maker(horizon, model)
    x, p = make_workload(st, model, batch=1, seed=600 + horizon)
    if maker is stairs_settings:
        place_on_step_flanks(x, st, seed=3)
    port, orc = AnalyticPort(st, model), Oracle(st, model)
    assert (port.n, port.m, port.nnz) == (orc.n, orc.m, orc.nnz)
    port.set_params(p[0])
    f, grad, g, jac = [np.copy(a) for a in port.eval(x[0], threads=1)]
    fo, grado, go, jaco = orc.eval(x[0], p[0])
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))  # noqa: E731
    assert rel(f, fo) < 1e-11 and rel(grad, grado) < 1e-11 and rel(g, go) < 1e-11 and rel(jac, jaco) < 1e-11
    f4, grad4, g4, jac4 = port.eval(x[0], threads=min(4, port.max_threads))
    assert f4 == f and np.array_equal(grad4, grad) and np.array_equal(g4, g) and np.array_equal(jac4, jac)
