"""The host-buffer callback path IPOPT binds (hipnlp_eval with host arrays), round 3: resident (doorbell) mode, auto-registered caller
arrays with their sentinel verification, `new_x` unknown.  Every variant must give the bits of the plain launch-per-call path."""
import numpy as np
import pytest

from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings
from hippopt_amd.synthetic import make_workload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def HipNlp():
    from hippopt_amd.hipnlp import HipNlp as cls
    return cls


def iterates(x, count, seed=3):
    rng = np.random.RandomState(seed)
    return [x + 1e-3 * i * rng.standard_normal(x.shape) for i in range(count)]


@pytest.mark.parametrize("maker,horizon,batch,lifted", [(periodic_step_settings, 100, 1, False), (periodic_step_settings, 100, 1, True),
                                                         (stairs_settings, 30, 2, False), (periodic_step_settings, 7, 5, True)])
def test_resident_mode_gives_the_launched_kernels_bits(model, HipNlp, maker, horizon, batch, lifted):
    """hipnlp_set_resident: the callback kernel waits on the device for a doorbell; outputs by system-scope stores; completion word.
    New x every call, every want mask IPOPT uses, cached requests in between; the kernel's idle exit and the session restart behind
    it; set_params in the middle (ends the session: the parameters are read afresh)."""
    import time
    st = maker(horizon, model)
    x, p = make_workload(st, model, batch=batch, seed=4200 + horizon)
    ref, res = HipNlp(st, model, batch=batch, detect_simple_bounds=lifted), HipNlp(st, model, batch=batch, detect_simple_bounds=lifted)
    for e in (ref, res):
        e.set_params(p)
        e.set_auto_register(False)
    res.set_resident(300.0)
    xs = iterates(x, 12)
    wants = [("f",), ("f", "g"), ("f", "grad", "g", "jac"), ("g",), ("jac",), ("grad",)]
    for i, xi in enumerate(xs):
        want = wants[i % len(wants)]
        a, b = ref.eval(xi, want=want), res.eval(xi, want=want)
        for u, v in zip(a, b):
            assert (u is None and v is None) or np.array_equal(u, v), (i, want)
        a, b = ref.eval(xi, new_x=False, want=("grad", "jac")), res.eval(xi, new_x=False, want=("grad", "jac"))   # cached, from HBM
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
        assert np.array_equal(ref.cost_terms()[1], res.cost_terms()[1])
        if i == 4:
            time.sleep(0.01)           # longer than the idle limit: the kernel has left, the next call starts another session
        if i == 7:
            p2 = p * (1.0 + 1e-3)
            ref.set_params(p2)
            res.set_params(p2)
    stats = res.host_stats()
    assert stats["resident_calls"] == len(xs) and stats["resident_sessions"] >= 3     # start, after the idle exit, after set_params
    # a non-finite evaluation is reported through the same flag
    bad = xs[0].copy()
    bad[0, 130:134] = 0.0              # zero base quaternion of knot 0: 1 / |q|
    f, grad, g, jac = res.eval(bad, nan_ok=True)
    with pytest.raises(Exception, match="non-finite"):
        res.eval(bad)
    assert not np.all(np.isfinite(g))
    res.set_resident(0.0)
    assert res.host_stats()["resident_alive"] == 0
    a, b = ref.eval(xs[1]), res.eval(xs[1])
    assert all(np.array_equal(u, v) for u, v in zip(a, b))


def test_resident_mode_refuses_launches_that_are_not_resident_at_once(model, HipNlp):
    from hippopt_amd.hipnlp import HipNlpError
    st = periodic_step_settings(100, model)
    eng = HipNlp(st, model, batch=8)
    with pytest.raises(HipNlpError) as e:
        eng.set_resident(100.0)
    assert e.value.code == -6


def test_caller_arrays_are_registered_at_their_second_sight_and_verified(model, HipNlp):
    """Auto-registration: the same output arrays twice in a row become direct kernel outputs (no staging copy), bitwise the same
    values; an array whose pages were replaced behind the library's back (munmap + mmap at the same address) is caught by the
    sentinel words, served through the pinned block and left alone from then on."""
    st = periodic_step_settings(100, model)
    x, p = make_workload(st, model, batch=1, seed=4300)
    ref, eng = HipNlp(st, model), HipNlp(st, model)
    for e in (ref, eng):
        e.set_params(p)
    ref.set_auto_register(False)
    xs = iterates(x, 6)
    out = (np.empty(1), np.empty((1, eng.n)), np.empty((1, eng.m)), np.empty((1, eng.nnz)))
    for i, xi in enumerate(xs):
        eng.eval(xi, out=out)
        r = ref.eval(xi)
        assert all(np.array_equal(u, v) for u, v in zip(out, r)), i
        assert eng.host_stats()["auto_ranges"] == (0 if i == 0 else 3)     # grad, g, jac: all >= 64 KB
    assert eng.host_stats()["auto_fallbacks"] == 0
    # IPOPT's order on registered arrays with early outputs: one evaluation fills every array
    eng.set_early_outputs(True)
    f_, grad_, g_, jac_ = out
    eng.eval(xs[2], new_x=None, want=("f",), out=(f_, None, None, None))
    eng.eval(xs[2], new_x=None, want=("g",), out=(None, None, g_, None))
    eng.eval(xs[2], new_x=None, want=("grad",), out=(None, grad_, None, None))
    eng.eval(xs[2], new_x=None, want=("jac",), out=(None, None, None, jac_))
    r = ref.eval(xs[2])
    assert all(np.array_equal(u, v) for u, v in zip(out, r))
    eng.set_early_outputs(False)

    # ---- stores that do not arrive in the caller's pages (what a registration gone stale would look like).  Not provoked by
    # unmapping memory under a live registration — a device write to an unmapped page is a GPU fault — but by the library's test
    # hook HIPNLP_DEBUG_MISDIRECT_AUTO: the array the handle registers is mapped to the pinned block's copy of the output instead
    import os
    eng2 = HipNlp(st, model)
    eng2.set_params(p)
    jac2 = np.empty((1, eng2.nnz))
    os.environ["HIPNLP_DEBUG_MISDIRECT_AUTO"] = "1"
    try:
        eng2.eval(xs[0], want=("jac",), out=(None, None, None, jac2))      # first sight
        eng2.eval(xs[1], want=("jac",), out=(None, None, None, jac2))      # second sight: registered (misdirected), verified, dropped
    finally:
        del os.environ["HIPNLP_DEBUG_MISDIRECT_AUTO"]
    assert np.array_equal(jac2, ref.eval(xs[1], want=("jac",))[3])          # right values all the same
    stats = eng2.host_stats()
    assert stats["auto_registered"] == 1 and stats["auto_fallbacks"] == 1 and stats["auto_ranges"] == 0
    for xi in xs[2:5]:                                                      # ... and the address is left alone from then on
        eng2.eval(xi, want=("jac",), out=(None, None, None, jac2))
        assert np.array_equal(jac2, ref.eval(xi, want=("jac",))[3])
    assert eng2.host_stats()["auto_ranges"] == 0 and eng2.host_stats()["auto_registered"] == 1
    # an explicit registration of an array the handle registered by itself takes it over (and its explicit unregistration ends it)
    eng.register_outputs(out)
    eng.eval(xs[4], out=out)
    assert all(np.array_equal(u, v) for u, v in zip(out, ref.eval(xs[4]))) and eng.host_stats()["auto_ranges"] == 0
    eng.unregister_outputs(out)
    eng.close()
    eng2.close()


def test_new_x_unknown_is_decided_by_comparison(model, HipNlp):
    st = periodic_step_settings(20, model)
    x, p = make_workload(st, model, batch=1, seed=4400)
    eng = HipNlp(st, model)
    eng.set_params(p)
    eng.set_host_timing(True)
    xs = iterates(x, 3)
    f0, *_ = eng.eval(xs[0], new_x=None, want=("f",))
    k0 = eng.last_kernel_ms()
    _, _, g0, _ = eng.eval(xs[0].copy(), new_x=None, want=("g",))     # equal values in another array: the cached evaluation
    f1, *_ = eng.eval(xs[1], new_x=None, want=("f",))                  # other values: evaluated
    ref = eng.eval(xs[0], new_x=True)
    assert f0[0] == ref[0][0] and np.array_equal(g0, ref[2]) and f1[0] != f0[0]
    assert k0 > 0.0
