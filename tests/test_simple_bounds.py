"""detect_simple_bounds (the reference runs Opti with {"expand": True, "detect_simple_bounds": True}, main_periodic_step.py:109-110):
the NLP the driver sees has the single-variable rows lifted into lbx / ubx.  The engine builds that reduced problem natively
(layout.h, `Layout::build(..., lift_simple_bounds)`; HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS).  Checked on the host emulation of the engine
(no GPU here; tests/test_gpu_parity.py repeats it on the device): sizes against SURVEY 8a (274 - 70 rows per interior knot), the
reduced g / jac g / pattern / bounds are exactly the kept rows of the full problem's, and the multipliers map back onto every named
constraint."""
import numpy as np

from hippopt_amd.hipnlp_solver import _SimpleBoundsLift
from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings, stairs_settings
from hippopt_amd.synthetic import make_workload

from emu_engine import EmuEngine  # noqa: E402


def test_reduced_problem_is_the_kept_rows_of_the_full_one(model):
    N = 6
    for maker in (periodic_step_settings, single_step_settings, stairs_settings):
        st = maker(N, model)
        x, p = make_workload(st, model, 1, 8)
        full = EmuEngine(st, model, p[0])
        red = EmuEngine(st, model, p[0], detect_simple_bounds=True)
        simple, var = full.simple_rows()
        kept, lb_full, ub_full = red.lift_map()
        keep_rows = np.nonzero(simple == 0)[0]
        assert np.array_equal(np.nonzero(kept >= 0)[0], keep_rows) and np.array_equal(kept[keep_rows], np.arange(keep_rows.size))
        # per interior knot: 24 u_v boxes + 23 joint position + 23 joint velocity boxes leave g (274 -> 204 rows)
        interior_full = sum(b[2] for b in full.row_blocks() if b[4] >= N - 1)
        per_knot_lifted = sum(b[2] for b in full.row_blocks() if b[4] >= N - 1 and simple[b[1]] == 1)
        assert (interior_full, per_knot_lifted) == (274, 70)
        x0_rows = 48 + 3 + 4 + 23 + 3
        fin_rows = 81 if st.final_state_expression_type == 1 else 0
        assert full.m - red.m == 70 * (N - 1) + 47 + x0_rows + fin_rows and red.m == keep_rows.size and red.n == full.n
        # bounds: kept rows keep theirs, a lifted box is a variable bound, an x_0 == initial_state row a fixed variable
        lbx, ubx, lbg, ubg = red.bounds()
        _, _, lbg_full, ubg_full = full.bounds()
        assert np.array_equal(lbg, lbg_full[keep_rows]) and np.array_equal(ubg, ubg_full[keep_rows])
        assert np.array_equal(lb_full, lbg_full) and np.array_equal(ub_full, ubg_full)
        ref_lbx, ref_ubx = np.full(full.n, -np.inf), np.full(full.n, np.inf)
        lifted = np.nonzero(simple)[0]
        np.maximum.at(ref_lbx, var[lifted], lbg_full[lifted])
        np.minimum.at(ref_ubx, var[lifted], ubg_full[lifted])
        assert np.array_equal(lbx, ref_lbx) and np.array_equal(ubx, ref_ubx) and np.sum(lbx == ubx) >= x0_rows
        # values and pattern: the full problem's with the lifted rows (one entry each) taken out
        f, grad, g, jac = red.eval(x)
        ff, gradf, gf, jacf = full.eval(x)
        ir, jc = full.sparsity()
        irr, jcr = red.sparsity()
        keep_entries = np.nonzero(simple[ir] == 0)[0]
        assert f[0] == ff[0] and np.array_equal(grad, gradf)
        assert np.array_equal(g[0], gf[0][keep_rows]) and np.array_equal(jac[0], jacf[0][keep_entries])
        assert irr.size == red.nnz == full.nnz - (full.m - red.m)    # every lifted row carried exactly one entry
        assert np.array_equal(keep_rows[irr], ir[keep_entries]) and np.array_equal(jcr, jc[keep_entries])
        assert np.all(np.diff(jcr.astype(np.int64) * (red.m + 1) + irr) > 0)   # still column-major, sorted


def test_multipliers_map_back_onto_every_named_constraint(model):
    N = 4
    st = periodic_step_settings(N, model)
    x, p = make_workload(st, model, 1, 9)
    eng = EmuEngine(st, model, p[0], detect_simple_bounds=True)
    lift = _SimpleBoundsLift(eng)
    rng = np.random.RandomState(0)
    lam_red, lam_x = rng.standard_normal(lift.m), rng.standard_normal(eng.n)
    lam = lift.full_multipliers(lam_red, lam_x)
    assert lam.size == eng.m_full and np.array_equal(lam[lift.keep_rows], lam_red)
    simple, var = eng.simple_rows()
    blocks = {b[0]: b for b in eng.row_blocks()}
    # a box row whose variable carries no other lifted row receives that variable's bound multiplier, whatever its sign
    first, rows, k0, nk = blocks["joint_velocity_bounds"][1:]
    r = first + rows * 2 + 5
    assert lam[r] == lam_x[var[r]]
    # the multiplier of a variable is given to exactly one of its lifted rows
    for v in np.unique(lift.lifted_vars):
        rows_v = lift.lifted_rows[lift.lifted_vars == v]
        assert np.count_nonzero(lam[rows_v]) <= 1 and (np.count_nonzero(lam[rows_v]) == 0 or lam[rows_v].sum() == lam_x[v])
    # Output.constraint_multipliers slices by named block (full numbering): every name is still there
    out = {name: lam[f:f + rws * nk_].reshape(nk_, rws) for name, f, rws, _k0, nk_ in eng.row_blocks()}
    assert "joint_velocity_bounds" in out and "joint_position_dynamics" in out and out["joint_velocity_bounds"].shape == (N, 23)
    # the snapshot of the lifted bounds follows the parameters (a changed initial state moves the x_0 rows' bounds)
    p2 = p[0].copy()
    p2[24 * N + 3:24 * N + 3 + 105] += 0.25      # initial_state block of the parameter vector
    eng.set_params(p2)
    lb_row, ub_row, _, _ = lift._lifted_bounds()
    _, lb_full, _ = eng.lift_map()
    assert np.array_equal(lb_row, lb_full[lift.lifted_rows])
