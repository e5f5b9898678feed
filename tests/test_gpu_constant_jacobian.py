"""The constant entries of jac g on the host path (hipnlp_set_constant_jacobian, HIPNLP_FLAG_JAC_VARYING_FIRST), through the C-ABI on
the GPU, against the CPU oracle: a host destination of the Jacobian is filled with the constants once per parameter set and the kernel
stores the varying entries only — what the caller reads must be the complete Jacobian, entry by entry, whichever way its array is fed
(pinned block + copy, registered by the handle, registered by the caller, fetched after the evaluation), after a parameter change, and
after the caller wrote over its own array."""
import gc

import numpy as np
import pytest

from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings
from hippopt_amd.synthetic import make_workload, place_on_step_flanks

pytestmark = pytest.mark.gpu
TOL = 1e-11


@pytest.fixture(scope="module")
def HipNlp():
    from hippopt_amd.hipnlp import HipNlp as cls
    return cls


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def iterates(x, count, seed=3):
    rng = np.random.RandomState(seed)
    return [x + 1e-3 * i * rng.standard_normal(x.shape) for i in range(count)]


def by_entry(eng, values):
    ir, jc = eng.sparsity()
    return {(int(r), int(c)): v for r, c, v in zip(ir, jc, values)}


def test_every_way_of_feeding_the_callers_array_gives_the_oracles_jacobian(model, HipNlp):
    from oracle_lib import Oracle
    st = periodic_step_settings(45, model)           # (grad f, g and jac g all reach the 64 KB the handle registers by itself)
    x, p = make_workload(st, model, batch=1, seed=7100)
    orc = Oracle(st, model)
    full = HipNlp(st, model)                      # every launch stores every entry: the behaviour before the switch existed
    full.set_params(p)
    full.set_constant_jacobian(False)
    xs = iterates(x, 6)
    want = [orc.eval(xi[0], p[0]) for xi in xs]
    mask = full.jac_constant_mask()
    assert 0.40 < mask.mean() < 0.46 and full.host_stats()["constant_entries"] == int(mask.sum())
    for vary_first in (False, True):
        eng = HipNlp(st, model, jac_varying_first=vary_first)
        eng.set_params(p)
        ref = by_entry(full, full.eval(xs[0])[3][0])
        assert by_entry(eng, eng.eval(xs[0])[3][0]) == ref                      # fresh arrays: pinned block + copy, bit for bit
        order = None
        if vary_first:
            ir, jc = eng.sparsity()
            pos = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(*full.sparsity()))}
            order = np.array([pos[(int(r), int(c))] for r, c in zip(ir, jc)])
        out = (np.empty(1), np.empty((1, eng.n)), np.empty((1, eng.m)), np.full((1, eng.nnz), np.nan))
        for i, xi in enumerate(xs):                                             # the same arrays again and again: registered at the second sight
            eng.eval(xi, out=out)
            fo, grado, go, jaco = want[i]
            jo = jaco if order is None else jaco[order]
            assert rel(out[0][0], fo) < TOL and rel(out[1][0], grado) < TOL and rel(out[2][0], go) < TOL and rel(out[3][0], jo) < TOL, (vary_first, i)
            fr = full.eval(xi)
            assert np.array_equal(out[3][0], fr[3][0] if order is None else fr[3][0][order]), (vary_first, i)
        stats = eng.host_stats()
        # (the constants are skipped in the varying-first order only: in CCS order the varying entries are fragments on the link)
        assert stats["auto_ranges"] == 3 and stats["constant_fills"] == int(vary_first) and stats["constant_refills"] == 0 and stats["auto_fallbacks"] == 0
        # the caller writes over its own (registered) array: the spot check sees it, the constants come back
        out[3][:] = 0.0
        eng.eval(xs[1], out=out)
        assert rel(out[3][0], want[1][3] if order is None else want[1][3][order]) < TOL
        assert eng.host_stats()["constant_refills"] == int(vary_first)
        # IPOPT's order with the Jacobian asked for AFTER its evaluation (new_x = 0): fetched from HBM into the registered array
        f_, grad_, g_, jac_ = out
        eng.eval(xs[3], new_x=True, want=("f",), out=(f_, None, None, None))
        jac_[:] = np.nan
        eng.eval(xs[3], new_x=False, want=("jac",), out=(None, None, None, jac_))
        assert rel(jac_[0], want[3][3] if order is None else want[3][3][order]) < TOL
        # arrays registered by the caller take the same route
        mine = (np.empty(1), np.empty((1, eng.n)), np.empty((1, eng.m)), np.full((1, eng.nnz), 7.0))
        eng.register_outputs(mine)
        try:
            for i in (4, 5):
                eng.eval(xs[i], out=mine)
                assert rel(mine[3][0], want[i][3] if order is None else want[i][3][order]) < TOL
        finally:
            eng.unregister_outputs(mine)
        # the zero-copy views of the pinned block are complete too
        views = eng.eval_pinned(xs[2])
        assert rel(views[3][0], want[2][3] if order is None else want[2][3][order]) < TOL
        eng.close()
    full.close()


def test_constants_follow_set_params(model, HipNlp):
    """dt and the mass sit in the constant entries: a new parameter set re-fills every destination the handle has filled before"""
    from oracle_lib import Oracle
    st = periodic_step_settings(12, model)
    x, p = make_workload(st, model, batch=1, seed=7200)
    p2 = p.copy()
    N = 12
    p2[0, 24 * N + 3 + 105 + 105] *= 1.6      # dt   (ParamOffsets, layout.h)
    p2[0, 24 * N] *= 0.85                     # mass
    orc = Oracle(st, model)
    eng = HipNlp(st, model, jac_varying_first=True)
    ir, jc = eng.sparsity()
    pos = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(*orc.sparsity()))}
    order = np.array([pos[(int(r), int(c))] for r, c in zip(ir, jc)])
    big = (np.empty(1), np.empty((1, eng.n)), np.empty((1, eng.m)), np.empty((1, eng.nnz)))
    eng.register_outputs(big)      # (at 12 knots the arrays are below the 64 KB the handle registers by itself)
    try:
        for params in (p, p2, p):
            eng.set_params(params)
            for xi in iterates(x, 3):
                eng.eval(xi, out=big)
                jo = orc.eval(xi[0], params[0])[3]
                assert rel(big[3][0], jo[order]) < TOL
            fresh = eng.eval(x)                       # unregistered: pinned block + copy
            assert rel(fresh[3][0], orc.eval(x[0], params[0])[3][order]) < TOL
        assert eng.host_stats()["constant_fills"] == 3 and eng.host_stats()["constant_refills"] == 0
    finally:
        eng.unregister_outputs(big)
    eng.close()


def test_batch_with_different_time_steps_and_the_stairs(model, HipNlp):
    """per-trajectory constants (every trajectory of a batch has its own dt), smooth terrain (HEIGHT / NORMAL entries are no longer
    constant there), both block orders, against a handle that stores every entry"""
    st = stairs_settings(9, model)
    B = 3
    x, p = make_workload(st, model, batch=B, seed=7300)
    place_on_step_flanks(x, st, seed=7300)
    N = 9
    for b in range(B):
        p[b, 24 * N + 3 + 105 + 105] *= 1.0 + 0.25 * b
    full = HipNlp(st, model, batch=B)
    full.set_params(p)
    full.set_constant_jacobian(False)
    want = full.eval(x)
    for vary_first in (False, True):
        eng = HipNlp(st, model, batch=B, jac_varying_first=vary_first)
        eng.set_params(p)
        out = tuple(np.full_like(w, np.nan) for w in want)
        eng.register_outputs(out)
        try:
            for _ in range(2):
                eng.eval(x, out=out)
                for b in range(B):
                    assert by_entry(eng, out[3][b]) == by_entry(full, want[3][b]), (vary_first, b)
                assert all(np.array_equal(a, w) for a, w in zip(out[:3], want[:3]))
        finally:
            eng.unregister_outputs(out)
        eng.close()
    full.close()


def test_two_live_handles_and_an_array_address_that_is_reused(model, HipNlp):
    """the table of registered ranges is process wide.  Handle A registers an array by itself; the array is freed and the
    allocator hands its address to a new array, which handle B (or A) is given: the registration is of pages that are gone, and the
    store must be verified whichever handle uses it — the values the caller reads are right, every time.  (Promoted from
    tools/diag/autoreg_stress.py.)"""
    st = periodic_step_settings(100, model)
    x, p = make_workload(st, model, batch=1, seed=7400)
    ref = HipNlp(st, model)
    ref.set_params(p)
    ref.set_auto_register(False)
    ref.set_constant_jacobian(False)
    xs = iterates(x, 4)
    want = [ref.eval(xi) for xi in xs]
    a, b = HipNlp(st, model), HipNlp(st, model)
    for e in (a, b):
        e.set_params(p)
    reused = fallbacks = 0
    for cycle in range(12):
        out = (np.empty(1), np.empty((1, a.n)), np.empty((1, a.m)), np.empty((1, a.nnz)))
        addr = out[3].ctypes.data
        for i in range(3):                           # A registers the three arrays at their second sight
            a.eval(xs[i], out=out)
            assert all(np.array_equal(u, v) for u, v in zip(out, want[i])), ("A", cycle, i)
        del out
        gc.collect()
        out2 = (np.empty(1), np.empty((1, a.n)), np.empty((1, a.m)), np.empty((1, a.nnz)))   # the allocator may hand the same addresses out again
        reused += int(out2[3].ctypes.data == addr)
        user = b if cycle % 2 == 0 else a
        for i in range(3):
            user.eval(xs[3 - i], out=out2)
            assert all(np.array_equal(u, v) for u, v in zip(out2, want[3 - i])), ("second owner", cycle, i)
        fallbacks = a.host_stats()["auto_fallbacks"] + b.host_stats()["auto_fallbacks"]
        del out2
        gc.collect()
    # (whether the allocator hands an address out again, and whether it does so with the same pages behind it, is its business: what is
    #  asserted is that the caller read the right values every time; the counters say what happened on this box)
    print("addresses reused: %d of 12, verified fallbacks: %d" % (reused, fallbacks))
    a.close()
    b.close()
    ref.close()


@pytest.mark.parametrize("vary_first", [True, False])
def test_device_resident_jacobian_holds_its_constants_and_is_repaired_when_overwritten(model, HipNlp, vary_first):
    """hipnlp_eval_device: the jac buffer is filled with the constants at its first sight (and after hipnlp_set_params), the launches
    store the varying entries of every block only (VARY instantiations: the constants are not even staged in LDS) — one run per block
    on a varying-first handle, scattered over the block on a handle in CasADi's CCS order — the buffer always holds the complete
    Jacobian; a buffer the caller wrote over is repaired by the kernel itself.  Reference: the CCS handle with
    hipnlp_set_constant_jacobian off (every entry computed, staged and stored by every launch).
    Both kernel variants (eight waves: one trajectory; four waves: a batch), both terrains, a shard handle too."""
    import torch
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    for maker, N, B, shard in ((periodic_step_settings, 40, 1, None), (periodic_step_settings, 12, 48, None), (stairs_settings, 10, 1, None),
                               (stairs_settings, 7, 80, None), (periodic_step_settings, 20, 1, (5, 13))):
        st = maker(N, model)
        x, p = make_workload(st, model, batch=B, seed=7500 + N)
        if maker is stairs_settings:
            place_on_step_flanks(x, st, seed=7500)
        kw = dict(knot_begin=shard[0], knot_end=shard[1]) if shard else {}
        full = HipNlp(st, model, batch=B, **kw)                              # (a CCS handle stores every entry unless asked)
        eng = HipNlp(st, model, batch=B, jac_varying_first=vary_first, **kw)
        if not vary_first:
            eng.set_constant_jacobian(True)                                  # CCS order, device destinations: opt-in
        ir, jc = eng.sparsity()
        pos = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(*full.sparsity()))}
        order = torch.from_numpy(np.array([pos[(int(r), int(c))] for r, c in zip(ir, jc)])).to(dev)
        if not vary_first:
            assert torch.equal(order, torch.arange(order.numel(), device=dev))
        p2 = p.copy()
        p2[:, 24 * N + 3 + 105 + 105] *= 1.3      # dt
        with torch.cuda.stream(stream):
            xd = [torch.from_numpy(xi).to(dev) for xi in iterates(x, 3)]
            # (a shard handle writes its own knots' part of every array: what lies outside must compare equal too)
            mk = lambda: [torch.full((B * k,), 0.0 if shard else float("nan"), dtype=torch.float64, device=dev) for k in (1, eng.n, eng.m, eng.nnz)]
            out, ref, other = mk(), mk(), mk()
        stream.synchronize()
        sh = stream.cuda_stream

        def both(i, dst):
            full.eval_device(xd[i].data_ptr(), *[t.data_ptr() for t in ref], stream=sh)
            eng.eval_device(xd[i].data_ptr(), *[t.data_ptr() for t in dst], stream=sh)
            stream.synchronize()
            for a, b_ in zip(dst[:3], ref[:3]):
                assert torch.equal(a, b_), (maker.__name__, N, B, i)
            got, want = dst[3].view(B, -1), ref[3].view(B, -1)[:, order]
            if shard:   # (a shard handle writes the blocks of its own knots: the rest of the array is not its business)
                mine = (torch.from_numpy(jc // 189).to(dev) >= shard[0]) & (torch.from_numpy(jc // 189).to(dev) < shard[1])
                got, want = got[:, mine], want[:, mine]
            assert torch.equal(got, want), (maker.__name__, N, B, i, int((got != want).sum()))
        for params in (p, p2):
            for e in (full, eng):
                e.set_params(params)
            for i in range(3):
                both(i, out)
            both(0, other)                      # a second buffer: filled at its first sight as well
        assert eng.host_stats()["constant_slices_healed"] == 0 and full.host_stats()["constant_fills"] == 0
        fills = eng.host_stats()["constant_fills"]
        assert fills == 4                       # two buffers x two parameter sets
        out[3].zero_()                          # the caller writes over its buffer between two calls ...
        both(1, out)                            # ... and still reads the complete Jacobian
        st_ = eng.host_stats()
        assert st_["constant_slices_healed"] > 0 and st_["constant_fills"] == fills
        both(2, out)
        assert eng.host_stats()["constant_slices_healed"] == st_["constant_slices_healed"]    # repaired once, in place from then on
        eng.set_constant_jacobian(False)        # every entry stored again: the same values
        out[3].fill_(0.0 if shard else float("nan"))
        both(0, out)
        eng.close()
        full.close()


def test_a_host_array_that_comes_back_under_a_new_registration_is_filled_again(model, HipNlp):
    """the record "this caller array holds the constants" used to be keyed by the bare address and the parameter generation.
    IPOPT frees its jac buffer, the next solve gets the same address (unchanged parameters), another user of the allocator wrote into
    the pages in between — anywhere but at the sixteen sampled entries: the launch stored the varying run only and the caller read wrong
    constants.  The record is now tied to the array's REGISTRATION: an array that was unregistered — by the public call, by
    hipnlp_host_release_auto_ranges, by a handle's eviction — is a new registration when it comes back, and is filled before it is trusted."""
    from oracle_lib import Oracle
    st = periodic_step_settings(40, model)
    x, p = make_workload(st, model, batch=1, seed=7700)
    xs = iterates(x, 6)
    orc = Oracle(st, model)
    eng = HipNlp(st, model, jac_varying_first=True)
    eng.set_params(p)
    full = HipNlp(st, model)
    full.set_params(p)
    mask = eng.jac_constant_mask()
    want = [by_entry(full, full.eval(xi, want=("jac",))[3][0]) for xi in xs]
    assert rel(np.array([want[0][k] for k in sorted(want[0])]), np.array([v for _, v in sorted(by_entry(full, orc.eval(xs[0][0], p[0])[3]).items())])) < TOL
    jac = np.empty((1, eng.nnz))

    def spoil():
        """every constant entry but the ones a spot check could look at first / last: what a second user of the pages leaves behind"""
        idx = np.flatnonzero(mask)
        jac[0, idx[7:-7:3]] = 12345.0

    # (a) registered by the caller, unregistered, spoiled, registered again
    for round_ in range(2):
        eng.register_outputs([jac])
        for i in range(2):
            eng.eval(xs[i + 2 * round_], want=("jac",), out=(None, None, None, jac))
            assert by_entry(eng, jac[0]) == want[i + 2 * round_], ("caller registration", round_, i)
        eng.unregister_outputs([jac])
        spoil()
    # (b) registered by the handle itself at the second sight, released by the process-wide call, spoiled, seen twice again
    for round_ in range(2):
        for i in range(3):
            eng.eval(xs[i + round_], want=("jac",), out=(None, None, None, jac))
            assert by_entry(eng, jac[0]) == want[i + round_], ("auto registration", round_, i)
        assert eng.host_stats()["auto_ranges"] >= 1
        eng.lib.hipnlp_host_release_auto_ranges()
        spoil()
    fills = eng.host_stats()["constant_fills"]
    assert fills >= 4, fills      # every fresh registration was filled (two by the caller, two by the handle)
    eng.close()
    full.close()


@pytest.mark.parametrize("vary_first", [True, False])
def test_device_buffer_sample_check_and_forget(model, HipNlp, vary_first):
    """the in-launch check of a device destination looks at four constants of every knot block (first, last, two between), not
    at the first alone; and hipnlp_forget_jac_destination drops the record of a buffer whose address now means other memory (a freed and
    re-allocated tensor): the next evaluation fills it again."""
    import torch
    dev = torch.device("cuda", 0)
    st = periodic_step_settings(30, model)
    B = 3
    x, p = make_workload(st, model, batch=B, seed=7800)
    full = HipNlp(st, model, batch=B)
    eng = HipNlp(st, model, batch=B, jac_varying_first=vary_first)
    if not vary_first:
        eng.set_constant_jacobian(True)
    for e in (full, eng):
        e.set_params(p)
    pos = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(*full.sparsity()))}
    order = torch.from_numpy(np.array([pos[(int(r), int(c))] for r, c in zip(*eng.sparsity())])).to(dev)
    mask = torch.from_numpy(eng.jac_constant_mask()).to(dev)
    xd = [torch.from_numpy(xi).to(dev) for xi in iterates(x, 3)]
    mk = lambda: [torch.full((B * k,), float("nan"), dtype=torch.float64, device=dev) for k in (1, eng.n, eng.m, eng.nnz)]
    out, ref = mk(), mk()

    def check(i):
        full.eval_device(xd[i].data_ptr(), *[t.data_ptr() for t in ref])
        eng.eval_device(xd[i].data_ptr(), *[t.data_ptr() for t in out])
        torch.cuda.synchronize()
        assert torch.equal(out[3].view(B, -1), ref[3].view(B, -1)[:, order]), i
    check(0)
    healed = eng.host_stats()["constant_slices_healed"]
    # the LAST constant of one interior knot block of the second trajectory (the old check looked at the first constant of a block only)
    jc = torch.from_numpy(eng.sparsity()[1]).to(dev)
    blk = (jc // 189) == 11
    last_const = int(torch.nonzero(blk & mask)[-1])
    out[3].view(B, -1)[1, last_const] = -7.0
    # (the four sampled positions of a block are looked at two per launch, alternating with the launch number: within two launches)
    for i in (1, 2):
        eng.eval_device(xd[i].data_ptr(), *[t.data_ptr() for t in out])
    torch.cuda.synchronize()
    check(1)
    assert eng.host_stats()["constant_slices_healed"] > healed
    # other memory at the same address: nothing the sample could see (every constant but the sampled ones), said with the forget call
    fills = eng.host_stats()["constant_fills"]
    out[3].fill_(float("nan"))
    assert eng.forget_jac_destination(out[3].data_ptr()) == 1
    check(2)
    assert eng.host_stats()["constant_fills"] == fills + 1
    assert eng.forget_jac_destination(0) == 1 and eng.forget_jac_destination(out[3].data_ptr()) == 0
    check(0)
    eng.close()
    full.close()
