"""The host-buffer callback path IPOPT binds (hipnlp_eval with host arrays), round 3: caller arrays registered by the handle at their
second sight with their sentinel verification, `new_x` unknown, early outputs in IPOPT's call order.  Every variant must give the bits
of the plain path (outputs through the pinned block)."""
import numpy as np
import pytest

from hippopt_amd.kinodyn_settings import periodic_step_settings
from hippopt_amd.synthetic import make_workload
from diag_lib import diag_library, diag_overrides

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def HipNlp():
    from hippopt_amd.hipnlp import HipNlp as cls
    return cls


def iterates(x, count, seed=3):
    rng = np.random.RandomState(seed)
    return [x + 1e-3 * i * rng.standard_normal(x.shape) for i in range(count)]


def test_caller_arrays_are_registered_at_their_second_sight_and_verified(model, HipNlp):
    """Auto-registration: the same output arrays twice in a row become direct kernel outputs (no staging copy), bitwise the same
    values; stores that do not arrive in the caller's pages are caught by the sentinel words, the call is served through the pinned
    block and the address is left alone from then on."""
    st = periodic_step_settings(100, model)
    x, p = make_workload(st, model, batch=1, seed=4300)
    ref, eng = HipNlp(st, model), HipNlp(st, model)
    for e in (ref, eng):
        e.set_params(p)
    ref.set_auto_register(False)
    ref.set_constant_jacobian(False)      # the reference stores every entry of jac g on every launch
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
    from diag_lib import diag_library, diag_overrides
    eng2 = HipNlp(st, model, library=diag_library())     # (the hook exists in the diagnostic build only)
    eng2.set_params(p)
    jac2 = np.empty((1, eng2.nnz))
    with diag_overrides(HIPNLP_DEBUG_MISDIRECT_AUTO=1):
        eng2.eval(xs[0], want=("jac",), out=(None, None, None, jac2))      # first sight
        eng2.eval(xs[1], want=("jac",), out=(None, None, None, jac2))      # second sight: registered (misdirected), verified, dropped
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


def test_hessian_with_ipopts_new_x_flag_uses_the_staged_x(model, HipNlp):
    """hipnlp_eval_hess_at: new_x = FALSE (IPOPT's eval_h at an accepted iterate: the x of the callbacks before it) skips the host copy
    of x and evaluates at the x the previous host-buffer call staged — the same bits as the plain call; with nothing staged the flag is
    ignored; `unknown` is decided by comparison"""
    st = periodic_step_settings(12, model)
    x, p = make_workload(st, model, batch=1, seed=4410)
    xs = iterates(x, 3)
    eng = HipNlp(st, model)
    eng.set_params(p)
    lam = np.random.RandomState(5).standard_normal((1, eng.m))
    ref = [eng.eval_hess(xi, 0.7, lam).copy() for xi in xs]
    assert not np.array_equal(ref[0], ref[1])
    fresh = HipNlp(st, model)
    fresh.set_params(p)
    assert np.array_equal(fresh.eval_hess(xs[1], 0.7, lam, new_x=False), ref[1])        # nothing staged yet: copied whatever the flag says
    fresh.eval(xs[2], want=("f", "g"))                                                   # IPOPT's sequence: callbacks at x, then eval_h(new_x = FALSE)
    assert np.array_equal(fresh.eval_hess(xs[2], 0.7, lam, new_x=False), ref[2])
    # a caller whose new_x = FALSE is wrong (the callbacks ran elsewhere, another iterate was evaluated in between): sixty-four words of x
    # are compared with the staged copy, a differing sample stages the argument — the Hessian of the x that was passed, not a stale one
    assert np.array_equal(fresh.eval_hess(xs[0], 0.7, lam, new_x=False), ref[0])
    assert np.array_equal(fresh.eval_hess(xs[0], 0.7, lam, new_x=False), ref[0])        # (... which is the staged one from then on)
    assert np.array_equal(fresh.eval_hess(xs[0], 0.7, lam, new_x=None), ref[0])         # unknown: compared, differs, copied
    assert np.array_equal(fresh.eval_hess(xs[0].copy(), 0.7, lam, new_x=None), ref[0])  # unknown: equal values in another array
    assert np.array_equal(fresh.eval_hess(xs[1], 0.7, lam), ref[1])                     # the plain call: new x
    # and the callbacks after a Hessian at a new x see that x as staged (new_x = False serves the cached evaluation only if there is one)
    f_ref = eng.eval(xs[1], want=("f",))[0][0]
    assert fresh.eval(xs[1], new_x=None, want=("f",))[0][0] == f_ref


def test_hessian_value_array_is_registered_at_its_second_sight_and_verified(model, HipNlp):
    """hipnlp_eval_hess: the caller's value array (IPOPT evaluates the Hessian into the value array of its own matrix) becomes a direct
    kernel output at its second consecutive sight, verified by the sentinel words like the callback outputs; the values are those of
    the path through the pinned block, bit for bit, and a store that does not arrive is caught and served through the pinned block."""
    import os
    st = periodic_step_settings(40, model)
    x, p = make_workload(st, model, batch=1, seed=4500)
    ref, eng = HipNlp(st, model), HipNlp(st, model)
    for e in (ref, eng):
        e.set_params(p)
    ref.set_auto_register(False)
    lam = np.random.RandomState(8).standard_normal((1, eng.m))
    xs = iterates(x, 5)
    out = np.empty((1, eng.hess_nnz()))
    for i, xi in enumerate(xs):
        got = eng.eval_hess(xi, 0.9, lam, out=out)
        assert got is out and np.array_equal(out, ref.eval_hess(xi, 0.9, lam)), i
        assert eng.host_stats()["auto_registered"] == (0 if i == 0 else 1)
    assert eng.host_stats()["auto_fallbacks"] == 0
    # the callback outputs of the same handle keep their own registrations beside it
    cb = (np.empty(1), np.empty((1, eng.n)), np.empty((1, eng.m)), np.empty((1, eng.nnz)))
    for xi in xs[:3]:
        eng.eval(xi, out=cb)
        assert all(np.array_equal(u, v) for u, v in zip(cb, ref.eval(xi)))
    assert np.array_equal(eng.eval_hess(xs[1], 0.9, lam, out=out), ref.eval_hess(xs[1], 0.9, lam))
    from diag_lib import diag_library, diag_overrides
    eng2 = HipNlp(st, model, library=diag_library())
    eng2.set_params(p)
    out2 = np.empty((1, eng2.hess_nnz()))
    with diag_overrides(HIPNLP_DEBUG_MISDIRECT_AUTO=1):
        eng2.eval_hess(xs[0], 0.9, lam, out=out2)
        eng2.eval_hess(xs[1], 0.9, lam, out=out2)      # registered (misdirected), caught, served through the pinned block
    assert np.array_equal(out2, ref.eval_hess(xs[1], 0.9, lam))
    stats = eng2.host_stats()
    assert stats["auto_registered"] == 1 and stats["auto_fallbacks"] == 1
    assert np.array_equal(eng2.eval_hess(xs[2], 0.9, lam, out=out2), ref.eval_hess(xs[2], 0.9, lam)) and eng2.host_stats()["auto_registered"] == 1
    eng.close()
    eng2.close()


def test_library_copies_survive_heap_memory_inside_a_stale_registration(model, HipNlp):
    """An array a handle registered by itself is freed while the registration is alive; the allocator hands the addresses out again —
    to the heap buffers of the library's own host-to-device copies (hipnlp_set_params stages 1 MB of parameter records for a batch of
    1024), which the HIP runtime then refuses with "invalid argument" because it still believes the old registration.  The library
    releases the self-registered ranges and retries (seen in bench.py: the throughput legs behind the Hessian's host-path timing)."""
    import gc
    st = periodic_step_settings(100, model)
    x, p = make_workload(st, model, batch=1, seed=4600)
    eng = HipNlp(st, model)
    eng.set_params(p)
    lam = np.random.RandomState(1).standard_normal((1, eng.m))
    for cycle in range(4):
        hv = np.empty((1, eng.hess_nnz()))               # ~1.2 MB: an mmap of its own
        for _ in range(3):
            eng.eval_hess(x, 1.0, lam, out=hv)           # registered by the handle at its second sight
        assert eng.host_stats()["auto_ranges"] >= 1
        del hv
        gc.collect()                                     # the array goes, its registration stays behind
        st2 = periodic_step_settings(10, model)
        x2, p2 = make_workload(st2, model, batch=1, seed=4601)
        B = 1200
        big = HipNlp(st2, model, batch=B)                # its set_params stages B x 936 B of GParams + B x 10 x 704 B of records on the heap
        big.set_params(np.tile(p2, (B, 1)))
        f, *_ = big.eval(np.tile(x2, (B, 1)), want=("f",))
        one = HipNlp(st2, model)
        one.set_params(p2)
        assert np.all(f == one.eval(x2, want=("f",))[0][0])
        big.close()
        one.close()
    eng.close()


def test_eval_remembers_addresses_per_array_object_not_per_id(model, HipNlp):
    """HipNlp.eval looks the data address of an argument array up once per array OBJECT (id + weak reference).  Arrays that die and
    whose ids are handed out again (CPython reuses them at once) must not inherit the dead array's address: the outputs of every call
    land in the arrays of that call, whatever happened to earlier ones; non-contiguous / non-float64 x still goes through a conversion."""
    import gc
    st = periodic_step_settings(12, model)
    x, p = make_workload(st, model, batch=1, seed=4700)
    eng, ref = HipNlp(st, model), HipNlp(st, model)
    for e in (eng, ref):
        e.set_params(p)
    ref.set_auto_register(False)
    want = ref.eval(x)
    ids = set()
    for cycle in range(6):
        out = (np.full(1, np.nan), np.full((1, eng.n), np.nan), np.full((1, eng.m), np.nan), np.full((1, eng.nnz), np.nan))
        ids.update(id(o) for o in out)
        xi = x.copy()
        got = eng.eval(xi, out=out)
        assert all(g is o for g, o in zip(got, out)) and all(np.array_equal(u, v) for u, v in zip(out, want)), cycle
        del out, got, xi
        gc.collect()
    assert len(ids) < 24       # ids were reused across the cycles (otherwise this test did not exercise what it is about)
    # x as a strided view and as float32: converted, same values up to the conversion
    wide = np.zeros((1, 2 * eng.n))
    wide[:, ::2] = x
    assert all(np.array_equal(u, v) for u, v in zip(eng.eval(wide[:, ::2]), want))
    f32 = eng.eval(x.astype(np.float32))
    assert np.array_equal(f32[0], ref.eval(x.astype(np.float32).astype(np.float64))[0])
    eng.close()
    ref.close()
    with pytest.raises(Exception):
        eng.eval(x)            # a closed handle is refused, not dereferenced


@pytest.mark.parametrize("terrain", ["planar", "stairs"])
def test_early_copy_out_of_the_host_path_changes_no_bit(model, HipNlp, terrain):
    """A varying-first handle's launches into host memory store the entries of jac g that are final after the second phase of the knot
    program behind that phase's barrier (written through the L2, so that they cross the link while the kinematic phases run) and leave
    them out of the copy-out at the end.  Which entries those are is RECORDED from the knot program (Layout::jslot_phase), not declared:
    a slot that got its value later than recorded would leave with a stale value.  Every output of every call must be bit for bit what
    the handle without the early pass gives (HIPNLP_EARLY_STORE=0, read by hipnlp_create) — plain arrays (pinned block) and arrays the
    handle has registered (direct stores), all four outputs and jac alone."""
    import os
    from hippopt_amd.kinodyn_settings import stairs_settings
    from hippopt_amd.synthetic import place_on_step_flanks
    st = (stairs_settings if terrain == "stairs" else periodic_step_settings)(45, model)
    x, p = make_workload(st, model, batch=1, seed=4800)
    if terrain == "stairs":
        place_on_step_flanks(x, st, seed=4800)
    engs = []
    for flag in ("1", "0"):
        with diag_overrides(HIPNLP_EARLY_STORE=flag) as lib:
            engs.append(HipNlp(st, model, jac_varying_first=True, library=lib))
        engs[-1].set_params(p)
    early, plain = engs
    outs = [(np.empty(1), np.empty((1, e.n)), np.empty((1, e.m)), np.empty((1, e.nnz))) for e in engs]
    xs = iterates(x, 5)
    for i, xi in enumerate(xs):
        for e, out in zip(engs, outs):
            e.eval(xi, out=out)                      # (first call: pinned block; from the second on: the registered arrays)
        assert all(np.array_equal(u, v) for u, v in zip(*outs)), i
    assert early.host_stats()["auto_ranges"] == plain.host_stats()["auto_ranges"] == 3
    for i, xi in enumerate(xs):                      # jac g alone, into the registered arrays and into fresh ones
        for e, out in zip(engs, outs):
            e.eval(xi + 1e-4, want=("jac",), out=(None, None, None, out[3]))
        assert np.array_equal(outs[0][3], outs[1][3]), i
        ja, jb = (e.eval(xi - 1e-4, want=("jac",))[3] for e in engs)
        assert np.array_equal(ja, jb), i
    for e in engs:
        e.close()


@pytest.mark.parametrize("terrain", ["planar", "stairs"])
def test_early_copy_out_of_the_hessian_host_path_changes_no_bit(model, HipNlp, terrain):
    """hipnlp_eval_hess through host buffers: the entries at the start of a knot's block — the point columns, a quarter (planar) to two fifths
    (smooth steps) of its bytes — are all emitted in the first phases of the Hessian program and leave behind that barrier, written
    through the L2, while the kinematic phases still run; the copy-out at the end skips them.  Which entries those are is RECORDED
    (HessLayout::pos_phase: every entry is emitted exactly once).  Bit for bit the handle without the early pass
    (hipnlp_set_hessian_early_run) and the device path, with fresh value arrays (pinned block) and a reused one (registered by the handle:
    direct stores); a batch of two and a shard handle too (the run of a shard's first knot starts where its values do)."""
    import torch
    from hippopt_amd.kinodyn_settings import stairs_settings
    from hippopt_amd.synthetic import place_on_step_flanks
    for B, kw in ((1, {}), (2, {}), (1, dict(knot_begin=7, knot_end=31))):
        st = (stairs_settings if terrain == "stairs" else periodic_step_settings)(45, model)
        x, p = make_workload(st, model, batch=B, seed=4850)
        if terrain == "stairs":
            place_on_step_flanks(x, st, seed=4850)
        engs = []
        for flag in (True, False):
            engs.append(HipNlp(st, model, batch=B, **kw))
            engs[-1].set_hessian_early_run(flag)
            assert engs[-1].hessian_early_run()["mode"] is flag and engs[-1].hessian_early_run()["in_use"] is flag
            engs[-1].set_params(p)
        dev_eng = HipNlp(st, model, batch=B, **kw)      # the product library's device path: its own staged launch, everything stored at the end
        dev_eng.set_params(p)
        rng = np.random.RandomState(9)
        lam, sig = rng.standard_normal((B, engs[0].m)), rng.uniform(0.3, 1.5, B)
        hn = engs[0].hess_nnz()
        outs = [np.full((B, hn), np.nan) for _ in engs]
        dev = torch.device("cuda", 0)
        ld, sd = torch.from_numpy(lam).to(dev), torch.from_numpy(sig).to(dev)
        for i, xi in enumerate(iterates(x, 4)):
            fresh = [e.eval_hess(xi, sig, lam) for e in engs]                       # fresh arrays: through the pinned block
            assert np.array_equal(fresh[0].view(np.int64), fresh[1].view(np.int64)), (terrain, B, kw, i)
            for e, o in zip(engs, outs):
                o.fill(np.nan)
                e.eval_hess(xi, sig, lam, out=o)                                    # the same array again and again: registered at its second sight
            assert np.array_equal(outs[0].view(np.int64), outs[1].view(np.int64)) and np.array_equal(outs[0].view(np.int64), fresh[0].view(np.int64)), (terrain, B, kw, i)
            hd = torch.full((B, hn), float("nan"), dtype=torch.float64, device=dev)
            xd = torch.from_numpy(xi).to(dev)
            dev_eng.eval_hess_device(xd.data_ptr(), sd.data_ptr(), ld.data_ptr(), hd.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(hd.cpu().numpy().view(np.int64), outs[0].view(np.int64)), (terrain, B, kw, i)
        assert engs[0].host_stats()["auto_fallbacks"] == 0
        for e in engs + [dev_eng]:
            e.close()


def test_hessian_early_run_is_decided_once_and_says_why(model, HipNlp):
    """Default mode of hipnlp_set_hessian_early_run: whether the run at the start of every knot block leaves early is faster is a property
    of the HOST — a gain with the calling thread on the card's NUMA node, a loss from the other socket — so the handle decides at its first
    Hessian call from what sysfs says (no clock: the same choice in every run); where the topology is not known it measures both ways on
    its first calls (three to warm up, nine of each kind, medians).  A handle whose launches cannot send a run ahead (long launches on the
    compact layout) decides nothing and says so.  The values are the same bits whatever it does; setting a mode starts it over."""
    from hippopt_amd.hipnlp import device_numa_node, parse_cpulist
    import os
    st = periodic_step_settings(30, model)
    x, p = make_workload(st, model, batch=1, seed=4870)
    eng, ref = HipNlp(st, model), HipNlp(st, model)
    for e in (eng, ref):
        e.set_params(p)
    ref.set_hessian_early_run(False)
    assert ref.hessian_early_run()["why"].startswith("set by the caller")
    lam = np.random.RandomState(2).standard_normal((1, eng.m))
    out, want = np.empty((1, eng.hess_nnz())), np.empty((1, eng.hess_nnz()))
    state = eng.hessian_early_run()
    assert (state["mode"], state["in_use"], state["us_off"], state["us_on"]) == (None, None, 0.0, 0.0) and state["why"].startswith("not decided yet")
    card = device_numa_node(0)
    mine = None
    if card is not None:
        cpu = os.sched_getaffinity(0)
        for node in range(64):
            try:
                cpus = parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())
            except OSError:
                break
            if cpu <= cpus:                       # (every CPU this thread may run on belongs to ONE node: the answer cannot change under the test)
                mine = node
    xs = iterates(x, 5)
    for i in range(24):
        eng.eval_hess(xs[i % 5], 0.9, lam, out=out)
        ref.eval_hess(xs[i % 5], 0.9, lam, out=want)
        assert np.array_equal(out, want), i
        state = eng.hessian_early_run()
        if card is not None and mine is not None:
            assert state["in_use"] is (mine == card), (i, state)          # from the first call on, without a clock
            assert state["us_off"] == 0.0 and state["us_on"] == 0.0 and "NUMA node" in state["why"]
    assert state["mode"] is None and state["in_use"] is not None           # decided by now either way (3 + 18 calls when measured)
    if "measured" in state["why"]:
        assert state["us_off"] > 0.0 and state["us_on"] > 0.0 and state["in_use"] == (state["us_on"] <= state["us_off"])
    eng.set_hessian_early_run(None)                                        # starts over
    assert eng.hessian_early_run()["in_use"] is None
    eng.close()
    ref.close()
    # a handle whose Hessian launches are long (compact layout: more than 512 workgroups) cannot send a run ahead: nothing to decide
    big = HipNlp(st, model, batch=20)
    xb, pb = make_workload(st, model, batch=20, seed=4871)
    big.set_params(pb)
    big.eval_hess(xb, 1.0, np.zeros((20, big.m)))
    state = big.hessian_early_run()
    assert state["in_use"] is False and state["why"].startswith("off: this handle's Hessian launches cannot send a run ahead")
    big.close()


def test_the_process_can_be_put_on_the_cards_side_of_the_host():
    """hipnlp_device_numa_node / pin_to_device_numa_node: the node is the one sysfs names for the card's PCI function, the calling thread ends
    up on CPUs of that node only (and only on CPUs it was allowed before); a host that does not say changes nothing"""
    import os
    from hippopt_amd.hipnlp import device_numa_node, parse_cpulist, pin_to_device_numa_node
    before = os.sched_getaffinity(0)
    try:
        node = device_numa_node(0)
        got = pin_to_device_numa_node(0)
        if node is None:
            assert got is None and os.sched_getaffinity(0) == before
        else:
            cpus = parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())
            if cpus & before:
                assert got == {"node": node, "cpus": len(cpus & before)} and os.sched_getaffinity(0) == cpus & before
            else:
                assert got is None and os.sched_getaffinity(0) == before
        # the C entry point (what a C / IPOPT caller uses) does the same to the calling thread
        import ctypes as C
        from hippopt_amd.hipnlp import load_library
        after_python = os.sched_getaffinity(0)
        os.sched_setaffinity(0, before)
        lib = load_library()
        lib.hipnlp_pin_thread_to_device_numa_node.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        n_out, c_out = C.c_int(-7), C.c_int(-7)
        assert lib.hipnlp_pin_thread_to_device_numa_node(0, C.byref(n_out), C.byref(c_out)) == 0
        assert os.sched_getaffinity(0) == after_python
        assert (n_out.value, c_out.value) == ((got["node"], got["cpus"]) if got else (-1, 0))
    finally:
        os.sched_setaffinity(0, before)


def test_arrays_the_wrapper_allocates_itself_are_never_page_locked(model, HipNlp):
    """HipNlp.eval / eval_hess without `out=` hand back fresh arrays the caller drops when it likes; the allocator would hand the same
    addresses out again on the next call and the handle would take that for "the same array twice in a row" and page-lock memory that
    is freed a moment later (a stale registration poisons every later HIP operation on host pointers inside it).  The wrapper holds the
    previous call's arrays until the next call has allocated its own: no registration ever, same values."""
    st = periodic_step_settings(9, model)
    x, p = make_workload(st, model, batch=1, seed=4900)
    eng = HipNlp(st, model)
    eng.set_params(p)
    lam = np.random.RandomState(2).standard_normal((1, eng.m))
    first, first_h = [a.copy() for a in eng.eval(x)], eng.eval_hess(x, 0.5, lam).copy()
    for i in range(6):
        got = [a.copy() for a in eng.eval(x)]           # the returned arrays die here, call after call
        assert all(np.array_equal(u, v) for u, v in zip(got, first)), i
        assert np.array_equal(eng.eval_hess(x, 0.5, lam), first_h), i
    stats = eng.host_stats()
    assert stats["auto_registered"] == 0 and stats["auto_ranges"] == 0
    eng.close()
