"""The constant entries of jac g (emit_jc in knot_body.h; Layout::jconst_pos) and the varying-first order of a knot's block
(HIPNLP_FLAG_JAC_VARYING_FIRST), on the host emulation against the AD oracle.

Which entries may be skipped by the host path is a STRUCTURAL statement of the kernel body (the trapezoid defects of
integrators/implicit_trapezoid.py:24-39, the x_0 rows of base/multiple_shooting_solver.py:713-742, the single-variable bound rows of
planner.py:386-405,699-719 are linear in x).  It is pinned here from both sides: an entry the mask calls constant has the SAME bits at
unrelated x and equals the oracle's entry there (soundness: nothing that varies is ever skipped); an entry the mask calls varying
does change between unrelated x (completeness: nothing is moved needlessly)."""
import numpy as np
import pytest

from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings, stairs_settings
from hippopt_amd.synthetic import make_workload, place_on_step_flanks
from hostemu_lib import HostEmu
from oracle_lib import Oracle


def _points(st, model, seeds):
    xs, p = [], None
    for s in seeds:
        x, pp = make_workload(st, model, 1, s)
        if getattr(st, "terrain_steps", None):
            place_on_step_flanks(x, st, seed=s)
        xs.append(x[0])
        p = pp[0] if p is None else p   # ONE parameter set, several x
    return xs, p


@pytest.mark.parametrize("maker", [periodic_step_settings, single_step_settings, stairs_settings])
@pytest.mark.parametrize("lifted", [False, True])
def test_constant_mask_is_sound_and_complete(model, maker, lifted):
    st = maker(5, model)
    e = HostEmu(st, model, detect_simple_bounds=lifted)
    mask = e.constant_mask()
    xs, p = _points(st, model, (11, 12, 13))
    jacs = [e.eval(x, p)[3] for x in xs]
    const = e.constant_fill(p)
    assert np.array_equal(np.isnan(const), ~mask)
    for j in jacs:   # soundness, bit for bit: what the library would fill in IS what the program computes, at every x
        assert np.array_equal(j[mask].view(np.uint64), const[mask].view(np.uint64))
    moved = np.zeros(mask.size, bool)
    for a in range(len(jacs)):
        for b in range(a + 1, len(jacs)):
            moved |= jacs[a] != jacs[b]
    assert not moved[mask].any()
    # (smooth terrain: d(row_x of the planar complementarity)/du_y = -y_x tau with y_x == 0 — a structural entry that is +-0.0 with the
    #  sign of tau's factor: it stays a varying entry, the sign bit of a zero is not worth a class of its own)
    zero = np.all([j == 0.0 for j in jacs], axis=0)
    assert (moved | zero)[~mask].all(), "entries that never change are being moved: %d" % int((~(moved | zero)[~mask]).sum())
    assert int(zero[~mask].sum()) <= (0 if maker is not stairs_settings else 8 * 5)
    if not lifted:   # the values of the oracle's forward-AD Jacobian (same pattern order: CCS; AD forms -dt/2 by another route: an ulp)
        o = Oracle(st, model)
        jo = o.eval(xs[0], p)[3]
        assert np.max(np.abs(jo[mask] - const[mask])) < 1e-15


def test_constant_share_at_the_bench_shape(model):
    """BASELINE config 4 (periodic, N = 100): 59 344 of 137 879 entries do not depend on x (VERDICT r03 counted them by evaluating the
    oracle at three x); 595 of the 1 386 entries of an interior knot"""
    st = periodic_step_settings(100, model)
    e = HostEmu(st, model)
    nvary, nnz_v, nconst = e.vary_counts()
    assert e.nnz == 137879 and nconst == 59344
    assert nnz_v[1] == 1386 and nnz_v[1] - nvary[1] == 595
    assert int(e.constant_mask().sum()) == nconst


def test_constants_follow_the_parameters(model):
    """dt and the mass are what the constants hold: another parameter set, other constants, same mask"""
    st = periodic_step_settings(4, model)
    e = HostEmu(st, model)
    x, p = make_workload(st, model, 1, 5)
    p2 = p[0].copy()
    po_dt = 24 * 4 + 3 + 105 + 105          # ParamOffsets::dt (layout.h)
    po_mass = 24 * 4
    assert p2[po_dt] == p[0][po_dt] and p2[po_dt] > 0
    p2[po_dt] *= 1.7
    p2[po_mass] *= 0.9
    c1, c2 = e.constant_fill(p[0]), e.constant_fill(p2)
    mask = e.constant_mask()
    assert np.array_equal(np.isnan(c2), ~mask)
    changed = c1[mask] != c2[mask]
    assert changed.any() and not changed.all()            # -dt/2 and mass entries follow, +-1 entries stay
    j2 = e.eval(x[0], p2)[3]
    assert np.array_equal(j2[mask].view(np.uint64), c2[mask].view(np.uint64))


@pytest.mark.parametrize("maker", [periodic_step_settings, single_step_settings, stairs_settings])
@pytest.mark.parametrize("lifted", [False, True])
def test_varying_first_order_is_a_permutation_inside_every_knot_block(model, maker, lifted):
    st = maker(4, model)
    ccs = HostEmu(st, model, detect_simple_bounds=lifted)
    vf = HostEmu(st, model, detect_simple_bounds=lifted, jac_varying_first=True)
    assert (ccs.n, ccs.m, ccs.nnz) == (vf.n, vf.m, vf.nnz)
    assert ccs.vary_counts() == vf.vary_counts()
    x, p = make_workload(st, model, 1, 21)
    if getattr(st, "terrain_steps", None):
        place_on_step_flanks(x, st, seed=21)
    j1, j2 = ccs.eval(x[0], p[0])[3], vf.eval(x[0], p[0])[3]
    r1, c1 = ccs.sparsity()
    r2, c2 = vf.sparsity()
    d1 = {(int(r), int(c)): v for r, c, v in zip(r1, c1, j1)}
    d2 = {(int(r), int(c)): v for r, c, v in zip(r2, c2, j2)}
    assert len(d1) == ccs.nnz and d1 == d2                 # same entries, same values (bit for bit)
    # block by block: the varying entries first in (col, row) order, then the constant ones in (col, row) order
    nvary, nnz_v, _ = vf.vary_counts()
    mask = vf.constant_mask()
    at = 0
    for k in range(4):
        v = 0 if k == 0 else (2 if k == 3 else 1)
        blk = slice(at, at + nnz_v[v])
        assert not mask[at:at + nvary[v]].any() and mask[at + nvary[v]:at + nnz_v[v]].all()
        assert set(c2[blk] // 189) == {k}
        for part in (slice(at, at + nvary[v]), slice(at + nvary[v], at + nnz_v[v])):
            key = list(zip(c2[part].tolist(), r2[part].tolist()))
            assert key == sorted(key)
        assert sorted(zip(c1[blk].tolist(), r1[blk].tolist())) == sorted(zip(c2[blk].tolist(), r2[blk].tolist()))
        at += nnz_v[v]
    assert mask[at:].all()                                  # the horizon-global entries
    # what the library fills in is consistent with the order
    fill = vf.constant_fill(p[0])
    assert np.array_equal(fill[mask].view(np.uint64), j2[mask].view(np.uint64))


@pytest.mark.parametrize("maker", [periodic_step_settings, single_step_settings, stairs_settings])
@pytest.mark.parametrize("lifted", [False, True])
def test_varying_slots_lie_inside_the_window_the_trimmed_scratch_stages(model, maker, lifted):
    """The four-wave VARY kernels stage jac slots [js::V0, js::V0 + js::vary_slots(terrain)) only, through a base moved back by js::V0: a
    recorded slot that depends on x outside that window would be a store outside the staging (in front of it, or into grad[]).  The layout
    derives the flag from what the RECORDER saw (hipnlp_create launches no VARY kernel when it is false): it must hold for every shipped
    settings combination, in both orders of a block, and the window must be used up to its last few slots (no stale slack)."""
    st = maker(6, model)
    for vf in (False, True):
        e = HostEmu(st, model, detect_simple_bounds=lifted, jac_varying_first=vf)
        ok, lo, hi, w0, w1 = e.vary_partition()
        assert ok and w0 <= lo <= hi < w1, (ok, lo, hi, w0, w1)
        assert w0 == 921
