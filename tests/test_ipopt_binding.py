"""IPOPT's C callbacks as compiled symbols (include/hipnlp_ipopt.h, hippopt_amd/csrc/hipnlp_ipopt.cpp): what nlpsol's IPOPT plugin
binds for the reference behind `opti.solver("ipopt", ...)` / `self._solver.solve()` (base/opti_solver.py:123-125, 479).

IPOPT is not in the image.  tests/ipopt_harness/harness.c holds IpStdCInterface.h's callback typedefs, assigns the library's
functions to them (a signature IPOPT could not bind does not compile) and replays IPOPT's call protocol; the values it records are
compared with the CPU oracle here."""
import json
import os
import struct
import subprocess

import ctypes as C
import numpy as np
import pytest

from hippopt_amd import _abi, hipnlp
from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings
from hippopt_amd.synthetic import make_workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS_SRC = os.path.join(ROOT, "tests", "ipopt_harness", "harness.c")
HARNESS = os.path.join(ROOT, "tests", "_build", "ipopt_harness")
TOL = 1e-11
KINDS = {0: "f", 1: "grad", 2: "g", 3: "jac", 4: "hess", 5: "jac_structure", 6: "hess_structure", 7: "bounds"}


def build_harness():
    lib_dir = os.path.dirname(hipnlp.library_path())
    deps = [HARNESS_SRC, hipnlp.library_path(), os.path.join(ROOT, "include", "hipnlp_ipopt.h"), os.path.join(ROOT, "include", "hipnlp.h")]
    if not os.path.exists(HARNESS) or any(os.path.getmtime(d) > os.path.getmtime(HARNESS) for d in deps):
        os.makedirs(os.path.dirname(HARNESS), exist_ok=True)
        subprocess.check_call(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=199309L", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                               HARNESS_SRC, "-L", lib_dir, "-lhipnlp", "-Wl,-rpath," + lib_dir, "-lm", "-o", HARNESS])
    return HARNESS


def test_ipopt_callback_signatures_bind_and_symbols_are_exported():
    """compiles the harness with -Werror: the library's functions are assigned to variables of IpStdCInterface.h's callback types"""
    hipnlp.load_library()
    build_harness()
    lib = C.CDLL(hipnlp.library_path())
    for name in ("hipnlp_ipopt_eval_f", "hipnlp_ipopt_eval_grad_f", "hipnlp_ipopt_eval_g", "hipnlp_ipopt_eval_jac_g", "hipnlp_ipopt_eval_h",
                 "hipnlp_ipopt_sizes", "hipnlp_ipopt_bounds", "hipnlp_ipopt_attach", "hipnlp_ipopt_detach"):
        assert hasattr(lib, name), name
    # a NULL handle is refused, not dereferenced
    lib.hipnlp_ipopt_eval_f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    obj = C.c_double()
    assert lib.hipnlp_ipopt_eval_f(5, None, 1, C.byref(obj), None) == 0


def write_input(path, desc, p, xs, lam, obj_factor, attach, shards=0):
    blob = C.string_at(C.addressof(desc), C.sizeof(desc))
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", 0x49504F54, len(blob), p.size, xs.shape[0], int(attach), int(shards)))
        f.write(blob)
        f.write(np.ascontiguousarray(p, np.float64).tobytes())
        f.write(np.ascontiguousarray(xs, np.float64).tobytes())
        f.write(np.ascontiguousarray(lam, np.float64).tobytes())
        f.write(struct.pack("<d", obj_factor))


def read_records(path):
    out = []
    raw = open(path, "rb").read()
    at = 0
    while at < len(raw):
        kind, point, ok, count = struct.unpack_from("<4i", raw, at)
        at += 16
        out.append((KINDS[kind], point, ok, np.frombuffer(raw, np.float64, count, at).copy()))
        at += 8 * count
    return out


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("maker,horizon,lifted,attach,vary_first", [(periodic_step_settings, 30, False, 0, False), (single_step_settings, 30, True, 1, True),
                                                                     (periodic_step_settings, 100, True, 1, True), (periodic_step_settings, 100, True, 2, True),
                                                                     (periodic_step_settings, 100, False, 2, False)])
def test_ipopt_protocol_replayed_in_c_matches_the_oracle(model, tmp_path, maker, horizon, lifted, attach, vary_first):
    """attach: 0 plain handle, 1 hipnlp_ipopt_attach (the harness fails if a callback at another x touches the g / jac buffers — the
    TNLPAdapter's caches), 2 + the opt-in early outputs.  vary_first: HIPNLP_FLAG_JAC_VARYING_FIRST, the order a C binding picks for
    IPOPT's triplets; the Jacobian values are compared entry by (row, column)."""
    from hess_util import hess_mismatch, triplets_to_dict
    from oracle_lib import Oracle
    st = maker(horizon, model)
    x, p = make_workload(st, model, batch=1, seed=5000 + horizon)
    rng = np.random.RandomState(2)
    points = 8
    xs = np.stack([x[0] + 1e-3 * i * rng.standard_normal(x[0].shape) for i in range(points)])
    xs[-1, 130:134] = 0.0                       # the last point: zero base quaternion of knot 0 -> NaN
    desc = _abi.DescC()
    desc.settings, desc.model, desc.batch = st.to_c(), model.to_c(), 1
    desc.flags = (_abi.FLAG_DETECT_SIMPLE_BOUNDS if lifted else 0) | (_abi.FLAG_JAC_VARYING_FIRST if vary_first else 0)
    orc = Oracle(st, model)
    # the reduced problem in terms of the oracle's full one
    is_simple = np.zeros(orc.m, np.int32)
    var = np.full(orc.m, -1, np.int32)
    if lifted:
        from hostemu_lib import HostEmu
        he = HostEmu(st, model)
        he.lib.hostemu_simple_rows(C.c_void_p(he.h), is_simple.ctypes.data_as(C.POINTER(C.c_int)), var.ctypes.data_as(C.POINTER(C.c_int)))
    keep_rows = np.nonzero(is_simple == 0)[0]
    iro, jco = orc.sparsity()
    keep_entries = np.nonzero(is_simple[iro] == 0)[0]
    new_row = np.full(orc.m, -1)
    new_row[keep_rows] = np.arange(keep_rows.size)
    lam = rng.standard_normal(keep_rows.size)
    lam_full = np.zeros(orc.m)
    lam_full[keep_rows] = lam
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_input(src, desc, p[0], xs, lam, 0.8, attach)
    res = subprocess.run([build_harness(), src, dst, "200"], capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr
    timing = json.loads(res.stdout.strip().splitlines()[-1])
    assert timing["ipopt_iterate_four_c_calls_us"] > 0
    if attach:
        assert timing["auto_registered"] >= 1                      # IPOPT's jac array (g too when it reaches 64 KB; the gradient the loop reuses)
        assert timing["constant_entries"] > 0 and (timing["constant_fills"] >= 1 or not vary_first) and timing["constant_refills"] == 0
    order = None   # position of the oracle's kept entry behind every entry of the handle's order
    recs = read_records(dst)
    seen = {k: 0 for k in KINDS.values()}
    ref = {}
    for kind, point, ok, v in recs:
        seen[kind] += 1
        if kind == "bounds":
            lbo, ubo = orc.bounds(p[0])
            n, m = orc.n, keep_rows.size
            assert ok == 1 and v.size == 2 * (n + m)
            gl, gu = v[2 * n:2 * n + m], v[2 * n + m:]
            assert np.array_equal(gl, np.clip(lbo[keep_rows], -2e19, 2e19)) and np.array_equal(gu, np.clip(ubo[keep_rows], -2e19, 2e19))
            xl, xu = v[:n], v[n:2 * n]
            if lifted:
                assert np.sum(np.abs(xl) < 1e19) >= is_simple.sum() // 2 and np.all(xl <= xu)
            else:
                assert np.all(xl == -2e19) and np.all(xu == 2e19)
            continue
        if kind == "jac_structure":
            assert ok == 1
            want_rc = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(new_row[iro[keep_entries]], jco[keep_entries]))}
            got_rc = list(zip(v[:v.size // 2].astype(int).tolist(), v[v.size // 2:].astype(int).tolist()))
            assert len(got_rc) == len(want_rc) == len(set(got_rc)) and all(rc in want_rc for rc in got_rc)
            order = np.array([want_rc[rc] for rc in got_rc])
            if not vary_first:
                assert np.array_equal(order, np.arange(order.size))       # CasADi's CCS order
            continue
        if kind == "hess_structure":
            assert ok == 1
            hrow, hcol = v[:v.size // 2].astype(int), v[v.size // 2:].astype(int)
            assert np.all(hrow >= hcol)
            continue
        if point == points - 1:                                    # the NaN point: FALSE, as IPOPT expects
            assert ok == 0, kind
            continue
        assert ok == 1, (kind, point)
        if point not in ref:
            ref[point] = orc.eval(xs[point], p[0])
        fo, grado, go, jaco = ref[point]
        if kind == "f":
            assert rel(v[0], fo) < TOL
        elif kind == "grad":
            assert rel(v, grado) < TOL
        elif kind == "g":
            assert rel(v, go[keep_rows]) < TOL
        elif kind == "jac":
            assert rel(v, jaco[keep_entries][order]) < TOL
        elif kind == "hess":
            err, where = hess_mismatch(triplets_to_dict(hrow, hcol, v), triplets_to_dict(*orc.hess(xs[point], p[0], 0.8, lam_full)))
            assert err < TOL, where
    # the protocol was really walked: accepted points asked for everything, rejected trial points for f and g only
    assert seen["f"] == points and seen["g"] == points and seen["grad"] == seen["jac"] and 0 < seen["grad"] < points and seen["hess"] >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("maker,horizon,lifted,attach", [(single_step_settings, 30, True, 1), (periodic_step_settings, 10, True, 2), (periodic_step_settings, 100, False, 1)])
def test_both_triplet_orders_hand_ipopt_the_same_matrix(model, tmp_path, maker, horizon, lifted, attach):
    """north_star asks for the reference's iterate sequence; the reference hands IPOPT CasADi's CCS triplet order (base/opti_solver.py:479 ->
    nlpsol), HipNlpSolver and the C binding of INTEGRATION.md create varying-first handles.  IPOPT reads jac g as a SET of triplets
    (it builds its own matrix from iRow / jCol once and copies values by position ever after): the same protocol replayed in C through
    both orders must deliver, at every point, bit for bit the same {(iRow, jCol): value} and the same f, grad f, g and Hessian values —
    then no solver behind the callbacks can tell the orders apart.  (What is NOT verified here: IPOPT / MUMPS themselves — absent.)"""
    st = maker(horizon, model)
    x, p = make_workload(st, model, batch=1, seed=5100 + horizon)
    rng = np.random.RandomState(4)
    points = 6
    xs = np.stack([x[0] + 1e-3 * i * rng.standard_normal(x[0].shape) for i in range(points)])
    recs = {}
    m_red = None
    for vary_first in (False, True):
        desc = _abi.DescC()
        desc.settings, desc.model, desc.batch = st.to_c(), model.to_c(), 1
        desc.flags = (_abi.FLAG_DETECT_SIMPLE_BOUNDS if lifted else 0) | (_abi.FLAG_JAC_VARYING_FIRST if vary_first else 0)
        if m_red is None:
            eng = hipnlp.HipNlp(st, model, detect_simple_bounds=lifted)
            m_red = eng.m
            eng.close()
            lam = rng.standard_normal(m_red)
        src, dst = str(tmp_path / ("in%d.bin" % vary_first)), str(tmp_path / ("out%d.bin" % vary_first))
        write_input(src, desc, p[0], xs, lam, 0.8, attach)
        res = subprocess.run([build_harness(), src, dst, "20"], capture_output=True, text=True, timeout=240)
        assert res.returncode == 0, res.stderr
        recs[vary_first] = read_records(dst)
    a, b = recs[False], recs[True]
    assert [(k, pt, ok) for k, pt, ok, _ in a] == [(k, pt, ok) for k, pt, ok, _ in b]       # the same walk through the protocol
    rc = {}
    for order, rs in recs.items():
        for kind, point, ok, v in rs:
            if kind == "jac_structure":
                rc[order] = list(zip(v[:v.size // 2].astype(int).tolist(), v[v.size // 2:].astype(int).tolist()))
    assert rc[False] != rc[True] and sorted(rc[False]) == sorted(rc[True]) and rc[False] == sorted(rc[False], key=lambda t: (t[1], t[0]))   # CCS = column major
    compared = 0
    for (kind, point, ok, va), (_, _, _, vb) in zip(a, b):
        if kind == "jac":
            da, db = dict(zip(rc[False], va.view(np.int64).tolist())), dict(zip(rc[True], vb.view(np.int64).tolist()))
            assert da == db, (point, sum(1 for k in da if da[k] != db[k]))
            compared += 1
        elif kind in ("f", "grad", "g", "hess", "bounds", "hess_structure"):
            assert np.array_equal(va.view(np.int64), vb.view(np.int64)), (kind, point)
    assert compared >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("maker,horizon,lifted,attach,vary_first,shards", [(periodic_step_settings, 100, True, 1, True, 4), (single_step_settings, 30, True, 2, True, 3),
                                                                            (periodic_step_settings, 30, False, 0, False, 2)])
def test_ipopt_protocol_through_a_multi_device_handle(model, tmp_path, maker, horizon, lifted, attach, vary_first, shards):
    """One caller, several shard handles (hipnlp_multi_create) behind the SAME five callbacks: IPOPT's protocol replayed in C against a
    handle of `shards` shards of device 0 records, at every call, the bits the plain handle records — structure, bounds, f, grad f, g,
    jac g, the Hessian values, and FALSE at the NaN point."""
    st = maker(horizon, model)
    x, p = make_workload(st, model, batch=1, seed=5200 + horizon)
    rng = np.random.RandomState(6)
    points = 7
    xs = np.stack([x[0] + 1e-3 * i * rng.standard_normal(x[0].shape) for i in range(points)])
    xs[-1, 189 * (horizon // 2) + 130:189 * (horizon // 2) + 134] = 0.0      # zero base quaternion in the middle of the horizon -> NaN
    desc = _abi.DescC()
    desc.settings, desc.model, desc.batch = st.to_c(), model.to_c(), 1
    desc.flags = (_abi.FLAG_DETECT_SIMPLE_BOUNDS if lifted else 0) | (_abi.FLAG_JAC_VARYING_FIRST if vary_first else 0)
    eng = hipnlp.HipNlp(st, model, detect_simple_bounds=lifted)
    lam = rng.standard_normal(eng.m)
    eng.close()
    recs, timing = {}, {}
    for k in (0, shards):
        src, dst = str(tmp_path / ("in%d.bin" % k)), str(tmp_path / ("out%d.bin" % k))
        write_input(src, desc, p[0], xs, lam, 0.8, attach, shards=k)
        res = subprocess.run([build_harness(), src, dst, "200"], capture_output=True, text=True, timeout=240)
        assert res.returncode == 0, res.stderr
        recs[k] = read_records(dst)
        timing[k] = json.loads(res.stdout.strip().splitlines()[-1])
    assert timing[shards]["shards"] == shards and timing[0]["shards"] == 0
    a, b = recs[0], recs[shards]
    assert [(k, pt, ok) for k, pt, ok, _ in a] == [(k, pt, ok) for k, pt, ok, _ in b]
    assert any(ok == 0 for _, pt, ok, _ in b if pt == points - 1)
    for (kind, point, ok, va), (_, _, _, vb) in zip(a, b):
        if point == points - 1:
            continue                                           # (the NaN point: which entries are NaN is the same, their payload bits need not be)
        assert np.array_equal(va.view(np.int64), vb.view(np.int64)), (kind, point)
    print("IPOPT iterate as four C calls, us: one handle %.1f, %d shards of one device %.1f" % (timing[0]["ipopt_iterate_four_c_calls_us"], shards, timing[shards]["ipopt_iterate_four_c_calls_us"]))
