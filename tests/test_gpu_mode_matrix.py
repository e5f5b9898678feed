"""One owner for the handle's mode matrix ("smaller" (ii)): what a handle does is the product of
    {order of a knot block: CCS | varying first} x {constants of jac g in place: off | on} x {where the outputs go: pinned block + copy,
    array registered by the handle itself, array registered by the caller, device buffer, peer buffer} x {detect_simple_bounds: off | on} x
    {whole horizon | knot shard}
— each pair is tested somewhere; here every cell of the table is walked on ONE small problem and compared, entry by (row, column) entry,
with the CPU oracle (1e-11), two iterates per cell so that the second evaluation finds the first one's destination (constants in place,
registrations made).  The reference side of the boundary is one function of x (nlpsol's nlp_f / nlp_g / nlp_grad_f / nlp_jac_g behind
/root/reference/src/hippopt/base/opti_solver.py:444-479): every cell must be that function."""
import itertools

import numpy as np
import pytest

from hippopt_amd.kinodyn_settings import periodic_step_settings
from hippopt_amd.synthetic import make_workload

pytestmark = pytest.mark.gpu
TOL = 1e-11
N = 36                      # (jac g of a whole-horizon handle reaches the 64 KB the handle registers by itself; a shard of 20 knots too)
SHARD = (9, 29)
DESTINATIONS = ("pinned block", "auto-registered", "caller-registered", "device", "peer")


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


@pytest.fixture(scope="module")
def problem(model):
    from hostemu_lib import HostEmu
    from oracle_lib import Oracle
    st = periodic_step_settings(N, model)
    x, p = make_workload(st, model, batch=1, seed=8800)
    rng = np.random.RandomState(5)
    xs = [x + 1e-3 * i * rng.standard_normal(x.shape) for i in range(2)]
    orc = Oracle(st, model)
    refs = [orc.eval(xi[0], p[0]) for xi in xs]
    iro, jco = orc.sparsity()
    he = HostEmu(st, model)
    import ctypes as C
    simple, var = np.zeros(orc.m, np.int32), np.zeros(orc.m, np.int32)
    he.lib.hostemu_simple_rows(C.c_void_p(he.h), simple.ctypes.data_as(C.POINTER(C.c_int)), var.ctypes.data_as(C.POINTER(C.c_int)))
    return st, p, xs, refs, iro, jco, simple


def expected(problem, i, lifted, shard):
    """the oracle's outputs at iterate i as {(row, col): value} / row and gradient slices of the cell"""
    st, p, xs, refs, iro, jco, simple = problem
    f, grad, g, jac = refs[i]
    keep_rows = np.nonzero(simple == 0)[0] if lifted else np.arange(g.size)
    new_row = np.full(g.size, -1)
    new_row[keep_rows] = np.arange(keep_rows.size)
    entries = {(int(new_row[r]), int(c)): v for r, c, v in zip(iro, jco, jac) if new_row[r] >= 0}
    return f, grad, g[keep_rows], entries


CELLS = [c for c in itertools.product((False, True), (False, True), DESTINATIONS, (False, True), (False, True))
         if not (c[2] == "peer" and c[1] and not c[0])]     # (peer stores of the varying runs need the varying-first order: hipnlp_eval_device_peers_vary)


@pytest.mark.parametrize("vary_first,constants,destination,lifted,shard", CELLS,
                         ids=["%s-%s-%s-%s-%s" % ("vf" if c[0] else "ccs", "const" if c[1] else "all", c[2].replace(" ", "_"), "lifted" if c[3] else "full", "shard" if c[4] else "whole") for c in CELLS])
def test_cell(model, problem, vary_first, constants, destination, lifted, shard):
    import torch
    from hippopt_amd.hipnlp import HipNlp
    st, p, xs, refs, iro, jco, simple = problem
    kw = dict(knot_begin=SHARD[0], knot_end=SHARD[1]) if shard else {}
    eng = HipNlp(st, model, detect_simple_bounds=lifted, jac_varying_first=vary_first, **kw)
    eng.set_params(p)
    eng.set_constant_jacobian(constants)
    ir, jc = eng.sparsity()
    kb, ke = SHARD if shard else (0, N)
    mine = (jc // 189 >= kb) & ((jc // 189 < ke) | ((jc >= 189 * N) & (ke == N)))      # entries of the handle's own knots (the global columns ride with the last knot)
    d = eng.dims
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    host_out = None
    if destination in ("auto-registered", "caller-registered"):
        host_out = (np.zeros(1), np.zeros((1, eng.n)), np.zeros((1, eng.m)), np.zeros((1, eng.nnz)))
        if destination == "caller-registered":
            eng.register_outputs(host_out)
    dev_out = [torch.zeros(k, dtype=torch.float64, device=dev) for k in (1, eng.n, eng.m, eng.nnz)] if destination == "device" else None
    tot = eng.n + eng.nnz + eng.m
    peer = torch.zeros(tot + 2, dtype=torch.float64, device=dev) if destination == "peer" else None
    table = torch.tensor([peer.data_ptr()], dtype=torch.int64, device=dev) if peer is not None else None
    try:
        for rep in range(3):            # iterates 0, 1, 0: the second and third evaluation find their destination as the previous one left it
            i = rep % 2
            f_want, grad_want, g_want, entries = expected(problem, i, lifted, shard)
            if destination == "pinned block":
                f, grad, g, jac = eng.eval(xs[i])
                f, grad, g, jac = f[0], grad[0], g[0], jac[0]
            elif host_out is not None:
                f, grad, g, jac = eng.eval(xs[i], out=host_out)
                f, grad, g, jac = f[0], grad[0], g[0], jac[0]
            elif destination == "device":
                xd = torch.from_numpy(xs[i]).to(dev)
                eng.eval_device(xd.data_ptr(), *[t.data_ptr() for t in dev_out], stream=stream.cuda_stream)
                stream.synchronize()
                f, grad, g, jac = (t.cpu().numpy() for t in dev_out)
                f = f[0]
            else:
                xd = torch.from_numpy(xs[i]).to(dev)
                if vary_first and constants:
                    eng.fill_jac_constants(peer.data_ptr() + 8 * eng.n, True, stream.cuda_stream)
                    eng.eval_device_peers_vary(xd.data_ptr(), table.data_ptr(), 1, 0, stream=stream.cuda_stream)
                else:
                    eng.eval_device_peers(xd.data_ptr(), table.data_ptr(), 1, 0, stream=stream.cuda_stream)
                stream.synchronize()
                o = peer.cpu().numpy()
                grad, jac, g, f = o[:eng.n], o[eng.n:eng.n + eng.nnz], o[eng.n + eng.nnz:tot], o[tot]
            cell = (vary_first, constants, destination, lifted, shard, rep)
            got = {(int(r), int(c)): v for r, c, v, own in zip(ir, jc, jac, mine) if own}
            # (a whole-horizon peer buffer filled by a shard handle holds the constants of EVERY knot: compared on the handle's own entries only)
            want_own = {rc: v for rc, v in entries.items() if (kb <= rc[1] // 189 < ke) or (rc[1] >= 189 * N and ke == N)}
            assert got.keys() == want_own.keys(), cell
            worst = max(abs(got[rc] - want_own[rc]) / max(1.0, abs(want_own[rc])) for rc in got)
            assert worst < TOL, (cell, worst)
            g0 = d.shard_grad_off
            assert rel(grad[g0:g0 + d.shard_grad], grad_want[g0:g0 + d.shard_grad]) < TOL, cell
            rows = np.unique(np.concatenate([r[r >= 0] for r in (eng.stage_rows(k) for k in range(kb, ke))]))
            assert rel(g[rows], g_want[rows]) < TOL, cell
            if not shard:
                assert rel(f, f_want) < TOL and rows.size == eng.m, cell
        stats = eng.host_stats()
        assert stats["auto_fallbacks"] == 0 and stats["constant_slices_healed"] == 0, (stats, destination)
        if destination == "auto-registered":
            assert stats["auto_ranges"] >= 1, stats
    finally:
        if destination == "caller-registered":
            eng.unregister_outputs(host_out)
        eng.close()
