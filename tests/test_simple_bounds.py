"""detect_simple_bounds (the reference runs Opti with {"expand": True, "detect_simple_bounds": True}, main_periodic_step.py:109-110):
the NLP the driver sees has the single-variable rows lifted into lbx / ubx.  Checked on the host emulation of the engine (no GPU
here): sizes against SURVEY 8a (274 - 70 rows per interior knot), the reduced g / jac g are the kept rows of the full ones, and the
multipliers map back onto every named constraint."""
import numpy as np

from hippopt_amd.hipnlp_solver import _SimpleBoundsLift
from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings
from hippopt_amd.synthetic import make_workload
from hostemu_lib import HostEmu


from emu_engine import EmuEngine  # noqa: E402


def test_reduced_problem_sizes_and_values(model):
    N = 6
    for maker in (periodic_step_settings, single_step_settings):
        st = maker(N, model)
        x, p = make_workload(st, model, 1, 8)
        eng = EmuEngine(st, model, p[0])
        lift = _SimpleBoundsLift(eng)
        blocks = {b[0]: b for b in eng.row_blocks()}
        # per interior knot: 24 u_v boxes + 23 joint position + 23 joint velocity boxes leave g (274 -> 204 rows)
        interior_full = sum(b[2] for b in eng.row_blocks() if b[4] >= N - 1)
        simple, var = eng.simple_rows()
        per_knot_lifted = sum(b[2] for b in eng.row_blocks() if b[4] >= N - 1 and simple[b[1]] == 1)
        assert (interior_full, per_knot_lifted) == (274, 70)
        x0_rows = 48 + 3 + 4 + 23 + 3
        fin_rows = 81 if st.final_state_expression_type == 1 else 0
        assert eng.m - lift.m == 70 * (N - 1) + 47 + x0_rows + fin_rows
        assert lift.n == eng.n
        lbx, ubx, lbg, ubg = lift.bounds()
        _, _, lbg_full, ubg_full = eng.bounds()
        assert np.array_equal(lbg, lbg_full[lift.keep_rows]) and lbg.size == lift.m
        # a lifted box is now a variable bound, an x_0 == initial_state row a fixed variable
        first, rows, k0, nk = blocks["joint_velocity_bounds"][1:]
        v = var[first]
        assert (lbx[v], ubx[v]) == (lbg_full[first], ubg_full[first]) and np.isfinite(lbx[v])
        first = blocks["joint_position_dynamics"][1] if "joint_position_dynamics[0]" not in blocks else None
        assert np.sum(lbx == ubx) >= x0_rows
        f, grad, g, jac = lift.eval(x)
        ff, gradf, gf, jacf = eng.eval(x)
        ir, jc = eng.sparsity()
        irr, jcr = lift.sparsity()
        assert np.array_equal(g[0], gf[0][lift.keep_rows]) and np.array_equal(jac[0], jacf[0][lift.keep_entries])
        assert irr.size == lift.nnz == eng.nnz - (eng.m - lift.m) + (24 if fin_rows else 0) * 0   # every lifted row carried exactly one entry
        assert np.array_equal(lift.keep_rows[irr], ir[lift.keep_entries]) and np.array_equal(jcr, jc[lift.keep_entries])
        # no kept row is a single plain variable any more, no column lost its bounds
        assert not np.any(simple[lift.keep_rows])


def test_multipliers_map_back_onto_every_named_constraint(model):
    N = 4
    st = periodic_step_settings(N, model)
    x, p = make_workload(st, model, 1, 9)
    eng = EmuEngine(st, model, p[0])
    lift = _SimpleBoundsLift(eng)
    lift.bounds()
    rng = np.random.RandomState(0)
    lam_red, lam_x = rng.standard_normal(lift.m), rng.standard_normal(eng.n)
    lam = lift.full_multipliers(lam_red, lam_x)
    assert lam.size == eng.m and np.array_equal(lam[lift.keep_rows], lam_red)
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
    # Output.constraint_multipliers slices by named block: every name is still there
    out = {name: lam[f:f + rws * nk_].reshape(nk_, rws) for name, f, rws, _k0, nk_ in eng.row_blocks()}
    assert "joint_velocity_bounds" in out and "joint_position_dynamics" in out and out["joint_velocity_bounds"].shape == (N, 23)
