"""One caller, several devices (hipnlp_multi_create): the ONE process that runs the NLP driver — the reference's
`self._solver.solve()`, /root/reference/src/hippopt/base/opti_solver.py:479 — hands x to a handle whose horizon is cut into shard
handles (contiguous knot ranges, defects owned by knot k + 1: base/multiple_shooting_solver.py:713-742) and gets f, grad f, g, jac g
and the Hessian values back in its own arrays.  The box has one card: the shards of these tests share device 0 (separate handles,
separate streams, the same addressing as on separate devices); what is compared is every output against the plain handle over the
whole horizon — bit for bit through host arrays — and against the CPU oracle."""
import numpy as np
import pytest

from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings, stairs_settings
from hippopt_amd.synthetic import make_workload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def HipNlp():
    from hippopt_amd.hipnlp import HipNlp as cls
    return cls


def iterates(x, count, seed=3):
    rng = np.random.RandomState(seed)
    return [x + 1e-3 * i * rng.standard_normal(x.shape) for i in range(count)]


def bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def same_kernel(plain_knots, batch, shards):
    """does every shard launch the instantiation of the knot kernel the plain handle launches (eight waves while the launch is resident
    at once: (knots + 1) x batch <= 512, hipnlp_create)?  Bit-identity is promised for that case; 1e-13 otherwise."""
    wide = lambda nk: nk <= 256 and (nk + 1) * batch <= 512  # noqa: E731
    return all(wide(s["knot_end"] - s["knot_begin"]) == wide(plain_knots) for s in shards)


@pytest.mark.parametrize("vary_first", [False, True])
@pytest.mark.parametrize("lifted", [False, True])
@pytest.mark.parametrize("maker,horizon", [(periodic_step_settings, 100), (single_step_settings, 30), (stairs_settings, 24)])
@pytest.mark.parametrize("n_shards", [2, 3, 4])
def test_shard_handles_on_one_device_give_the_plain_handles_bits(model, HipNlp, n_shards, maker, horizon, lifted, vary_first):
    st = maker(horizon, model)
    x, p = make_workload(st, model, batch=1, seed=7100 + horizon)
    kw = dict(detect_simple_bounds=lifted, jac_varying_first=vary_first)
    plain, multi = HipNlp(st, model, **kw), HipNlp(st, model, devices=[0] * n_shards, **kw)
    shards = multi.shards()
    assert [s["device"] for s in shards] == [0] * n_shards and shards[0]["knot_begin"] == 0 and shards[-1]["knot_end"] == horizon
    assert all(a["knot_end"] == b["knot_begin"] for a, b in zip(shards, shards[1:]))
    sizes = [s["knot_end"] - s["knot_begin"] for s in shards]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert plain.shards() == []
    assert (multi.n, multi.m, multi.nnz) == (plain.n, plain.m, plain.nnz)
    for e in (plain, multi):
        e.set_params(p)
    assert all(np.array_equal(u, v) for u, v in zip(plain.bounds(), multi.bounds()))
    assert all(np.array_equal(u, v) for u, v in zip(plain.sparsity(), multi.sparsity()))
    assert all(np.array_equal(u, v) for u, v in zip(plain.hess_sparsity(), multi.hess_sparsity()))
    assert same_kernel(horizon, 1, shards)
    lam = np.random.RandomState(11).standard_normal((1, plain.m))
    out = (np.empty(1), np.empty((1, multi.n)), np.empty((1, multi.m)), np.empty((1, multi.nnz)))
    hess = np.empty((1, multi.hess_nnz()))
    for i, xi in enumerate(iterates(x, 5)):
        ref = plain.eval(xi)
        got = multi.eval(xi, out=out)                     # (the same arrays every call: registered at their second sight, stored into directly)
        for name, u, v in zip(("f", "grad", "g", "jac"), got, ref):
            assert bits_equal(u, v), (name, i)
        assert bits_equal(multi.cost_terms()[1], plain.cost_terms()[1])
        hr = plain.eval_hess(xi, 0.7, lam)
        assert bits_equal(multi.eval_hess(xi, 0.7, lam, out=hess), hr), i
        assert bits_equal(multi.eval_hess(xi, 0.7, lam, out=hess, new_x=False), hr), i
    stats = multi.host_stats()
    big = sum(a.nbytes >= 64 * 1024 for a in out[1:] + (hess,))              # (arrays of at least 64 KB are registered at their second sight)
    assert stats["auto_ranges"] == big >= 2 and stats["auto_fallbacks"] == 0
    if vary_first:
        assert stats["constant_fills"] >= 1
    # one launching thread per shard: the same bits
    multi.set_threads(True, spin_us=50.0)
    for i, xi in enumerate(iterates(x, 4, seed=8)):
        ref = plain.eval(xi)
        got = multi.eval(xi, out=out)
        assert all(bits_equal(u, v) for u, v in zip(got, ref)), i
        assert bits_equal(multi.eval_hess(xi, 0.7, lam, out=hess), plain.eval_hess(xi, 0.7, lam)), i
        if i == 1:
            import time
            time.sleep(0.01)                              # (the workers go to sleep: the next call wakes them)
    if n_shards == 3:
        multi.set_threads(False)
    # IPOPT's order with new_x = FALSE behind the first callback: one evaluation, outputs that stayed in the shards' HBM are fetched
    xi = iterates(x, 7)[6]
    ref = plain.eval(xi)
    multi.set_prefetch(("f", "g"))
    before = multi.host_stats()["evaluations"]
    f_, *_ = multi.eval(xi, new_x=True, want=("f",))
    _, _, g_, _ = multi.eval(xi, new_x=False, want=("g",))
    _, grad_, _, _ = multi.eval(xi, new_x=False, want=("grad",))
    _, _, _, jac_ = multi.eval(xi, new_x=False, want=("jac",))
    assert multi.host_stats()["evaluations"] == before + 1
    for u, v in zip((f_, grad_, g_, jac_), ref):
        assert bits_equal(u, v)
    # g left in HBM (scattered over the constraint blocks) and asked for later: evaluated again, same values
    multi.set_prefetch(("f",))
    multi.eval(xi, new_x=True, want=("f",))
    _, _, g_, _ = multi.eval(xi, new_x=False, want=("g",))
    assert bits_equal(g_, ref[2]) and multi.host_stats()["evaluations"] == before + 3
    plain.close()
    multi.close()


def test_multi_handle_matches_the_oracle_and_flags_non_finite_values(model, HipNlp):
    from oracle_lib import Oracle
    st = periodic_step_settings(30, model)
    x, p = make_workload(st, model, batch=1, seed=7200)
    multi = HipNlp(st, model, devices=[0, 0, 0])
    multi.set_params(p)
    orc = Oracle(st, model)
    f, grad, g, jac = multi.eval(x)
    fo, grado, go, jaco = orc.eval(x[0], p[0])
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(1.0, np.abs(np.asarray(b)))))  # noqa: E731
    assert rel(f[0], fo) < 1e-11 and rel(grad[0], grado) < 1e-11 and rel(g[0], go) < 1e-11 and rel(jac[0], jaco) < 1e-11
    bad = x.copy()
    bad[0, 189 * 12 + 130:189 * 12 + 134] = 0.0        # zero base quaternion at knot 12 (the second shard): NaN
    from hippopt_amd.hipnlp import HipNlpError
    with pytest.raises(HipNlpError) as err:
        multi.eval(bad)
    assert err.value.code == -5
    out = multi.eval(bad, nan_ok=True)
    assert not np.all(np.isfinite(out[2]))
    f2, *_ = multi.eval(x, want=("f",))                 # ... and the next evaluation is clean again
    assert f2[0] == f[0]
    # the device-resident calls belong to one device
    with pytest.raises(HipNlpError) as err:
        multi.eval_device(8)
    assert err.value.code == -6
    multi.close()


@pytest.mark.parametrize("batch,n_shards", [(4, 3), (18, 2)])
def test_batched_multi_handle(model, HipNlp, batch, n_shards):
    """batch > 1 (config 5's shape, shortened): x is [batch][n]; big batches read x from each shard's own HBM (its knots' records, the
    halo, the globals, the other horizon end) instead of the pinned block"""
    st = stairs_settings(40, model)
    x, p = make_workload(st, model, batch=batch, seed=7300)
    plain, multi = HipNlp(st, model, batch=batch, jac_varying_first=True), HipNlp(st, model, batch=batch, jac_varying_first=True, devices=[0] * n_shards)
    for e in (plain, multi):
        e.set_params(p)
    exact = same_kernel(40, batch, multi.shards())
    lam = np.random.RandomState(2).standard_normal((batch, plain.m))
    sig = np.linspace(0.5, 1.0, batch)
    for xi in iterates(x, 3):
        ref, got = plain.eval(xi), multi.eval(xi)
        for name, u, v in zip(("f", "grad", "g", "jac"), got, ref):
            assert bits_equal(u, v) if exact else np.allclose(u, v, rtol=1e-13, atol=1e-13), name
        hr, hg = plain.eval_hess(xi, sig, lam), multi.eval_hess(xi, sig, lam)
        assert bits_equal(hg, hr) if exact else np.allclose(hg, hr, rtol=1e-12, atol=1e-12)
    plain.close()
    multi.close()


def test_parameters_changed_between_calls_reach_every_shard(model, HipNlp):
    st = periodic_step_settings(20, model)
    x, p = make_workload(st, model, batch=1, seed=7400)
    plain, multi = HipNlp(st, model, jac_varying_first=True), HipNlp(st, model, jac_varying_first=True, devices=[0, 0])
    out = (np.empty(1), np.empty((1, multi.n)), np.empty((1, multi.m)), np.empty((1, multi.nnz)))
    p2 = p.copy()
    from hippopt_amd.kinodyn_layout import ParamLayout
    p2[0, ParamLayout(20).dt] *= 1.5                    # dt: the constant entries of jac g change with it
    for q in (p, p2, p):
        plain.set_params(q)
        multi.set_params(q)
        for xi in iterates(x, 3):
            ref = plain.eval(xi)
            got = multi.eval(xi, out=out)
            assert all(bits_equal(u, v) for u, v in zip(got, ref))
    plain.close()
    multi.close()


def test_solver_on_two_shards_walks_the_single_device_iterates(model):
    """HipNlpSolver(devices=[0, 0]) through the planner's own wiring: the NLP driver of this process drives both shards and walks the
    iterates of the single handle — f, grad f, g, jac g are the same bits, so is everything the driver derives from them"""
    from test_gpu_solver_order import solve
    numeric = single_step_settings(30, model)
    (out_1, trace_1, rc_1, info_1), (out_2, trace_2, rc_2, info_2) = (solve(model, numeric, True, 61, 12, devices=d) for d in (None, [0, 0]))
    assert rc_1 == rc_2 and len(trace_1) == len(trace_2) >= 11
    for (it_1, x_1, f_1, pr_1), (it_2, x_2, f_2, pr_2) in zip(trace_1, trace_2):
        assert it_1 == it_2 and np.array_equal(x_1, x_2) and f_1 == f_2 and pr_1 == pr_2
    assert info_1["callbacks"] == info_2["callbacks"] and out_1.cost_value == out_2.cost_value
    assert out_1.cost_values == out_2.cost_values
