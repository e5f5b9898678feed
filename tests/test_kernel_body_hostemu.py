"""The kernel's knot program (hippopt_amd/csrc/knot_body.h) and the layout tables, run through the
TEST-ONLY host emulation (tests/hostemu), against the AD oracle.  This checks the hand-derived analytic
Jacobians and the CCS bookkeeping on a machine without a GPU; the GPU parity tests proper are in
test_gpu_parity.py."""
import numpy as np
import pytest

from hippopt_amd import _abi
from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings, stairs_settings
from hippopt_amd.synthetic import make_workload, place_on_step_flanks
from hostemu_lib import HostEmu
from oracle_lib import Oracle

TOL = 1e-11


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


@pytest.mark.parametrize("maker", [periodic_step_settings, single_step_settings])
@pytest.mark.parametrize("horizon", [2, 3, 6])
def test_body_matches_oracle(model, maker, horizon):
    st = maker(horizon, model)
    o, e = Oracle(st, model), HostEmu(st, model)
    assert (o.n, o.m, o.nnz) == (e.n, e.m, e.nnz)
    assert o.row_blocks() == e.row_blocks()
    ir, jc = o.sparsity()
    ir2, jc2 = e.sparsity()
    assert np.array_equal(ir, ir2) and np.array_equal(jc, jc2)
    x, p = make_workload(st, model, 1, 100 + horizon)
    f, grad, g, jac = o.eval(x[0], p[0])
    f2, grad2, g2, jac2, ct = e.eval(x[0], p[0])
    assert not np.isnan(g2).any() and not np.isnan(jac2).any()   # every row / entry is written by exactly one knot
    assert rel(np.array(f2), np.array(f)) < TOL and rel(grad2, grad) < TOL and rel(g2, g) < TOL and rel(jac2, jac) < TOL
    assert np.allclose(ct, o.cost_terms(), rtol=1e-12, atol=1e-10)
    lb, ub = o.bounds(p[0])
    lb2, ub2 = e.bounds(p[0])
    assert np.array_equal(lb, lb2) and np.array_equal(ub, ub2)


@pytest.mark.parametrize("horizon", [2, 5])
@pytest.mark.parametrize("oriented", [False, True])
def test_smooth_terrain_body_matches_oracle(model, horizon, oriented):
    """Stairs configuration (main_walking_on_stairs.py): closed-form third-order terrain jets of the kernel against the nested
    forward-mode oracle, with the points ON the flanks of the bumps and far away from them (exp underflow branch)."""
    st = stairs_settings(horizon, model)
    if oriented:
        st.terrain_steps[0]["orientation"] = 0.4
        st.terrain_steps[1]["orientation"] = -1.1
        st.terrain_steps[1]["position"] = (0.8, 0.2, 0.03)
        st.terrain_steps[1]["edge_sharpness"], st.terrain_steps[1]["side_sharpness"] = 3, 4
    o, e = Oracle(st, model), HostEmu(st, model)
    assert (o.n, o.m, o.nnz) == (e.n, e.m, e.nnz)
    assert o.row_blocks() == e.row_blocks()
    ir, jc = o.sparsity()
    ir2, jc2 = e.sparsity()
    assert np.array_equal(ir, ir2) and np.array_equal(jc, jc2)
    for flank in (True, False):
        x, p = make_workload(st, model, 1, 300 + horizon)
        if flank:
            place_on_step_flanks(x, st, seed=horizon)
        f, grad, g, jac = o.eval(x[0], p[0])
        f2, grad2, g2, jac2, ct = e.eval(x[0], p[0])
        assert not np.isnan(g2).any() and not np.isnan(jac2).any()
        assert rel(np.array(f2), np.array(f)) < 1e-9 and rel(grad2, grad) < 1e-9 and rel(g2, g) < TOL and rel(jac2, jac) < 1e-9
        assert np.allclose(ct, o.cost_terms(), rtol=1e-10, atol=1e-9)


def test_smooth_terrain_body_matches_reference_planner_fixture(model):
    """The kernel body (host emulation) directly against the stairs golden vectors of the reference's planner code."""
    import json
    import os
    from test_golden_planner import GOLD, settings_for
    z = np.load(os.path.join(GOLD, "planner_stairs_N3.npz"))
    st = settings_for(json.loads(str(z["meta"])), model)
    e = HostEmu(st, model)
    f, grad, g, jac, _ = e.eval(z["x"], z["p"])
    assert rel(g, z["g"]) < TOL and rel(grad, z["grad"]) < 1e-10 and abs(f - float(z["f"])) < 1e-10 * max(1.0, abs(float(z["f"])))
    ir, jc = e.sparsity()
    J = {(int(r), int(c)): v for r, c, v in zip(ir, jc, jac)}
    for r, c, v in zip(z["jac_row"], z["jac_col"], z["jac_val"]):
        got = J.pop((int(r), int(c)), None)
        assert (abs(v) < 1e-12) if got is None else (abs(got - v) <= 1e-10 * max(1.0, abs(v)))
    assert max((abs(v) for v in J.values()), default=0.0) < 1e-12


def test_cost_modes(model):
    st = periodic_step_settings(4, model)
    st.final_state_expression_type = _abi.EXPR_MINIMIZE
    st.periodicity_expression_type = _abi.EXPR_MINIMIZE
    st.final_state_expression_weight, st.periodicity_expression_weight = 2.5, 0.3
    st.joint_reg_as_coded = False
    st.contacts_centroid_cost_multiplier = 100.0
    o, e = Oracle(st, model), HostEmu(st, model)
    x, p = make_workload(st, model, 1, 21)
    f, grad, g, jac = o.eval(x[0], p[0])
    f2, grad2, g2, jac2, ct = e.eval(x[0], p[0])
    assert (o.m, o.nnz) == (e.m, e.nnz)
    assert rel(np.array(f2), np.array(f)) < TOL and rel(grad2, grad) < TOL and rel(g2, g) < TOL and rel(jac2, jac) < TOL


def test_far_from_origin_base_position(model):
    """Base-centred kinematics: accuracy must not degrade with |p_b| (N = 800 knots walk 8 m)."""
    st = single_step_settings(3, model)
    x, p = make_workload(st, model, 1, 22)
    for k in range(3):
        x[0][189 * k + 127:189 * k + 130] += [50.0, -30.0, 0.0]
        for c in range(8):
            x[0][189 * k + 15 * c + 6:189 * k + 15 * c + 9] += [50.0, -30.0, 0.0]
        x[0][189 * k + 180:189 * k + 183] += [50.0, -30.0, 0.0]
    o, e = Oracle(st, model), HostEmu(st, model)
    f, grad, g, jac = o.eval(x[0], p[0])
    f2, grad2, g2, jac2, ct = e.eval(x[0], p[0])
    assert rel(g2, g) < 1e-10 and rel(jac2, jac) < 1e-10 and rel(grad2, grad) < 1e-10


def test_invalid_models_are_rejected(model):
    import copy
    bad = copy.deepcopy(model)
    bad.parent = bad.parent.copy()
    bad.parent[5] = 7  # link 6 under link 7, which hangs under link 6: a cycle, not a tree rooted at link 0
    with pytest.raises(RuntimeError):
        HostEmu(periodic_step_settings(3, bad), bad)
    bad3 = copy.deepcopy(model)
    bad3.parent = bad3.parent.copy()
    bad3.parent[4] = 5  # a joint whose parent link is its own child link
    with pytest.raises(RuntimeError):
        HostEmu(periodic_step_settings(3, bad3), bad3)
    bad2 = copy.deepcopy(model)
    bad2.frame_link = bad2.frame_link.copy()
    bad2.frame_link[0] = 15  # sole on the knee link: 4-joint chain, the kernel wants 6
    with pytest.raises(RuntimeError):
        HostEmu(periodic_step_settings(3, bad2), bad2)


def renumbered(model, perm):
    """the same robot with joint j renamed perm[j] (child link j + 1 -> perm[j] + 1): what a joints_name_list that does not follow
    the kinematic tree produces (the x layout is the list order, variables.py:182-217)"""
    import copy
    m = copy.deepcopy(model)
    nj = len(perm)
    link = np.concatenate([[0], np.asarray(perm) + 1])   # old link -> new link
    for j in range(nj):
        jn = perm[j]
        m.parent[jn] = link[model.parent[j]]
        m.R_fix[jn], m.o_fix[jn], m.axis[jn] = model.R_fix[j], model.o_fix[j], model.axis[j]
        m.min_joint_positions[jn], m.max_joint_positions[jn] = model.min_joint_positions[j], model.max_joint_positions[j]
    for l in range(nj + 1):
        m.mass[link[l]], m.com[link[l]], m.inertia[link[l]] = model.mass[l], model.com[l], model.inertia[l]
    m.frame_link = np.array([link[l] for l in model.frame_link], np.int32)
    m.joint_names = [None] * nj
    for j in range(nj):
        m.joint_names[perm[j]] = model.joint_names[j]
    return m


@pytest.mark.parametrize("seed", [0, 1])
def test_joint_numbering_need_not_follow_the_tree(model, seed):
    """ergoCub's joints_name_list names torso_pitch before torso_roll whatever the URDF chains first: the engine takes ANY numbering
    of a tree rooted at link 0.  A randomly renumbered robot against the oracle (callback quartet and exact Hessian), and the cost
    against the same robot in tree order at the same physical state."""
    from hess_util import hess_mismatch, triplets_to_dict
    perm = np.random.RandomState(seed).permutation(23)
    m2 = renumbered(model, perm)
    assert any(int(m2.parent[j]) > j + 1 for j in range(23))   # really not topological
    for maker, N in ((periodic_step_settings, 4), (stairs_settings, 3)):
        st2 = maker(N, m2)
        st2.joint_regularization_cost_weights = np.linspace(0.5, 2.0, 23)
        x2, p2 = make_workload(st2, m2, 1, 900 + seed)
        o, he = Oracle(st2, m2), HostEmu(st2, m2)
        fo, grado, go, jaco = o.eval(x2[0], p2[0])
        f, grad, g, jac, _ = he.eval(x2[0], p2[0])
        iro, jco = o.sparsity()
        ir, jc = he.sparsity()
        assert np.array_equal(ir, iro) and np.array_equal(jc, jco)
        assert rel(f, fo) < 1e-11 and rel(grad, grado) < 1e-11 and rel(g, go) < 1e-11 and rel(jac, jaco) < 1e-11
        lam = np.random.RandomState(seed).standard_normal(o.m)
        ref = triplets_to_dict(*o.hess(x2[0], p2[0], 0.7, lam))
        hr, hc = he.hess_sparsity()
        err, where = hess_mismatch(triplets_to_dict(hr, hc, he.hess(x2[0], p2[0], 0.7, lam)), ref, diag_scaled=maker is stairs_settings)
        assert err < (1e-9 if maker is stairs_settings else 1e-11), where
    # the same physical state in tree order gives the same cost
    from hippopt_amd import kinodyn_layout as KL
    st1, st2 = periodic_step_settings(4, model), periodic_step_settings(4, m2)
    st2.joint_regularization_cost_weights = np.linspace(0.5, 2.0, 23)
    st1.joint_regularization_cost_weights = st2.joint_regularization_cost_weights[perm]
    x2, p2 = make_workload(st2, m2, 1, 77)
    x1, p1 = x2.copy(), p2.copy()
    pl = KL.ParamLayout(4)
    joint_blocks_x = [189 * k + off for k in range(4) for off in (134, 157)]
    joint_blocks_p = [pl.ref(k) + pl.REF["joint_regularization"] for k in range(4)] + [pl.init + 79, pl.fin + 79, pl.jpmax, pl.jpmin, pl.jvmax, pl.jvmin]
    for b in joint_blocks_x:
        x1[0, b:b + 23] = x2[0, b + perm]
    for b in joint_blocks_p:
        p1[0, b:b + 23] = p2[0, b + perm]
    f2, _, g2, _, _ = HostEmu(st2, m2).eval(x2[0], p2[0])
    f1, _, g1, _, _ = HostEmu(st1, model).eval(x1[0], p1[0])
    assert abs(f1 - f2) <= 1e-10 * max(1.0, abs(f2))
    assert abs(np.sort(g1) - np.sort(g2)).max() <= 1e-10 * max(1.0, np.abs(g2).max())   # the rows are a permutation of each other


# ---- exact Hessian of the Lagrangian (knot_hess_body.h) -------------------------------------------------------------------------
def hess_case(model, st, seed, sigma=0.8):
    from hess_util import hess_mismatch, triplets_to_dict
    x, p = make_workload(st, model, 1, seed)
    x, p = x[0], p[0]
    o, he = Oracle(st, model), HostEmu(st, model)
    lam = np.random.RandomState(seed).standard_normal(o.m)
    ref = triplets_to_dict(*o.hess(x, p, sigma, lam))
    ir, jc = he.hess_sparsity()
    mine = triplets_to_dict(ir, jc, he.hess(x, p, sigma, lam))
    return hess_mismatch(mine, ref), ir, jc, st


@pytest.mark.parametrize("maker", [periodic_step_settings, single_step_settings])
@pytest.mark.parametrize("horizon", [2, 3, 5])
def test_hessian_body_matches_oracle(model, maker, horizon):
    (err, where), ir, jc, st = hess_case(model, maker(horizon, model), 700 + horizon)
    assert err <= TOL, where
    # block diagonal by knot, the same block pattern at every knot
    assert np.all(ir // 189 == jc // 189)
    nk = ir.size // horizon
    assert ir.size == nk * horizon
    for k in range(1, horizon):
        assert np.array_equal(ir[k * nk:(k + 1) * nk] - 189 * k, ir[:nk]) and np.array_equal(jc[k * nk:(k + 1) * nk] - 189 * k, jc[:nk])


def test_hessian_body_cost_modes(model):
    """horizon-end expressions as costs (first / last coupling), the joint cost as intended, other weights"""
    st = periodic_step_settings(3, model)
    st.final_state_expression_type = st.periodicity_expression_type = _abi.EXPR_MINIMIZE
    st.final_state_expression_weight, st.periodicity_expression_weight = 2.0, 0.5
    st.contacts_centroid_cost_multiplier = 100.0
    (err, where), ir, jc, _ = hess_case(model, st, 711)
    assert err <= TOL, where
    off = ir // 189 != jc // 189
    assert off.sum() == 84 and np.all(ir[off] // 189 == 2) and np.all(jc[off] // 189 == 0)
    st = single_step_settings(3, model)
    st.joint_reg_as_coded = not st.joint_reg_as_coded
    (err, where), *_ = hess_case(model, st, 712, sigma=1.0)
    assert err <= TOL, where


@pytest.mark.parametrize("oriented", [False, True])
def test_hessian_body_smooth_terrain(model, oriented):
    """Stairs configuration: truncated Taylor polynomials of the terrain (fourth-order Z, knot_hess_terrain.h) against the oracle's
    nested forward AD, points ON the flanks of the bumps and far from them.  The terrain exponents (10, 20) make single entries
    of the point blocks as large as 1e7-1e8 next to O(1) ones: tolerance 1e-9 entrywise, as for the smooth-terrain Jacobian."""
    from hess_util import hess_mismatch, triplets_to_dict
    st = stairs_settings(3, model)
    if oriented:
        st.terrain_steps[0]["orientation"] = 0.4
        st.terrain_steps[1]["orientation"] = -1.1
        st.terrain_steps[1]["position"] = (0.8, 0.2, 0.03)
        st.terrain_steps[1]["edge_sharpness"], st.terrain_steps[1]["side_sharpness"] = 3, 4
    o, he = Oracle(st, model), HostEmu(st, model)
    ir, jc = he.hess_sparsity()
    for flank in (True, False):
        x, p = make_workload(st, model, 1, 303)
        if flank:
            place_on_step_flanks(x, st, seed=3)
        lam = np.random.RandomState(3).standard_normal(o.m)
        ref = triplets_to_dict(*o.hess(x[0], p[0], 0.8, lam))
        vals = he.hess(x[0], p[0], 0.8, lam)
        assert not np.isnan(vals).any()
        err, where = hess_mismatch(triplets_to_dict(ir, jc, vals), ref, diag_scaled=True)
        assert err <= 1e-9, where
        if flank:
            assert max(abs(v) for v in ref.values()) > 1e5   # the flanks are really exercised


@pytest.mark.parametrize("name", ["planner_periodic_N3", "planner_single_N3", "planner_costends_N2", "planner_stairs_N3", "planner_ramp_N3"])
def test_hessian_body_matches_reference_planner_fixture(model, name):
    import json
    import os
    from test_golden_planner import GOLD, hessian_times, settings_for
    z = np.load(os.path.join(GOLD, name + ".npz"))
    if "hess_dirs" not in z.files:
        pytest.skip("fixture without Hessian-vector products")
    he = HostEmu(settings_for(json.loads(str(z["meta"])), model), model)
    ir, jc = he.hess_sparsity()
    vals = he.hess(z["x"], z["p"], float(z["hess_sigma"]), z["hess_lambda"])
    assert rel(hessian_times(ir, jc, vals, he.n, z["hess_dirs"]), z["hess_times_dirs"]) <= (1e-9 if ("stairs" in name or "ramp" in name) else TOL)


@pytest.mark.parametrize("mode", ["subject_to", "minimize", "mixed", "single", "stairs", "stairs-minimize"])
def test_compact_scratch_layout_gives_the_same_values(model, mode):
    """The planar device kernel runs on a compact scratch (four workgroups per CU): own[] on top of the joint records, the horizon-end
    g rows inside the end-term partials, joint frames and link inertials read from the global tables.  Emulated in program order on the host:
    bitwise the values of the full layout, in every combination of the horizon-end expression types."""
    N = 5
    st = single_step_settings(N, model) if mode == "single" else (stairs_settings(N, model) if mode.startswith("stairs") else periodic_step_settings(N, model))
    if mode in ("minimize", "stairs-minimize"):
        st.final_state_expression_type = st.periodicity_expression_type = _abi.EXPR_MINIMIZE
        st.final_state_expression_weight, st.periodicity_expression_weight = 2.0, 0.5
    if mode == "mixed":
        st.periodicity_expression_type = _abi.EXPR_MINIMIZE
        st.periodicity_expression_weight = 0.25
    x, p = make_workload(st, model, 1, 700)
    if mode.startswith("stairs"):
        place_on_step_flanks(x, st, seed=2)
    he = HostEmu(st, model)
    full = he.eval(x[0], p[0])
    comp = he.eval(x[0], p[0], compact=True)
    for a, b in zip(full, comp):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    fo, grado, go, jaco = Oracle(st, model).eval(x[0], p[0])
    assert rel(np.array(comp[0]), np.array(fo)) < TOL and rel(comp[2], go) < TOL and rel(comp[3], jaco) < TOL and rel(comp[1], grado) < TOL


@pytest.mark.parametrize("terrain", ["planar", "stairs"])
@pytest.mark.parametrize("waves", [4, 8])
def test_groups_of_one_phase_on_different_waves_do_not_depend_on_each_other(model, terrain, waves):
    """On the GPU the task groups of one phase run concurrently on different waves; only the groups of ONE wave are ordered (table order).
    The host expansions run everything in table order and cannot see a dependency between two waves of a phase (a task group moved to
    another wave of its phase once passed every CPU test and failed on the GPU).  Here every phase's groups run wave by wave in three
    different orders of the waves — for both columns of the program table, both terrains, all variants of the horizon ends: the results
    must be the table-order results, bit for bit (also catches two waves adding into one scratch entry: the order would show in the
    last bits)."""
    from hippopt_amd.kinodyn_settings import stairs_settings
    from hippopt_amd.synthetic import place_on_step_flanks
    N = 4
    st = (stairs_settings if terrain == "stairs" else periodic_step_settings)(N, model)
    st.final_state_expression_type = st.periodicity_expression_type = _abi.EXPR_MINIMIZE
    x, p = make_workload(st, model, batch=1, seed=321)
    if terrain == "stairs":
        place_on_step_flanks(x, st, seed=2)
    e = HostEmu(st, model)
    ref = e.eval(x[0], p[0])
    for order in (0, 1, 2):
        got = e.eval_wave_order(x[0], p[0], waves, order)
        assert got[0] == ref[0], (order, got[0], ref[0])
        for a, b, name in zip(got[1:], ref[1:], ("grad", "g", "jac", "cost terms")):
            assert np.array_equal(a, b), (order, name, int(np.argmax(a != b)))
        # the entries of jac g recorded as final after the second phase (what a launch into host memory stores early) ARE final there,
        # whatever the order of the waves — in the EIGHT-wave column of the program table: only those kernels store early (hipnlp.hip
        # EARLY_OUT); the four-wave kernel of the planar terrain runs some first-phase groups later (knot_body.h sched4p, round 6)
        if waves == 8 or terrain == "stairs":
            assert e.early_violations() == 0, (order, e.early_violations())
    jp, gp = e.output_phases()
    assert jp.max() <= 5 and (jp <= 1).sum() > 0.25 * jp.size and gp.max() <= 5     # every entry has a phase; a good part is early


@pytest.mark.parametrize("terrain", ["planar", "stairs"])
def test_hessian_program_phases_are_wave_order_independent(model, terrain):
    """the same check for the Hessian program of the knot (four waves): every phase wave by wave in three orders of the waves"""
    import hostemu_lib
    st = (stairs_settings if terrain == "stairs" else periodic_step_settings)(3, model)
    st.final_state_expression_type = st.periodicity_expression_type = _abi.EXPR_MINIMIZE
    x, p = make_workload(st, model, batch=1, seed=322)
    if terrain == "stairs":
        place_on_step_flanks(x, st, seed=3)
    e = HostEmu(st, model)
    lam = np.random.RandomState(6).standard_normal(e.m)
    ref = e.hess(x[0], p[0], 0.8, lam)
    try:
        for order in (0, 1, 2):
            hostemu_lib.set_wave_order(order)
            got = e.hess(x[0], p[0], 0.8, lam)
            assert np.array_equal(got, ref), (order, int(np.argmax(got != ref)))
    finally:
        hostemu_lib.set_wave_order(-1)


@pytest.mark.parametrize("terrain", ["planar", "stairs"])
def test_compact_layout_phases_are_wave_order_independent(model, terrain):
    """the four-wave column on the COMPACT scratch (arrays with disjoint lifetimes share storage): two waves of one phase touching
    storage that is shared in this layout would show as an order dependence"""
    import hostemu_lib
    st = (stairs_settings if terrain == "stairs" else periodic_step_settings)(4, model)
    st.final_state_expression_type = st.periodicity_expression_type = _abi.EXPR_SUBJECT_TO
    x, p = make_workload(st, model, batch=1, seed=323)
    if terrain == "stairs":
        place_on_step_flanks(x, st, seed=4)
    e = HostEmu(st, model)
    ref = e.eval(x[0], p[0], compact=True)
    try:
        for order in (0, 1, 2):
            hostemu_lib.set_wave_order(order)
            got = e.eval(x[0], p[0], compact=True)
            assert got[0] == ref[0]
            for a, b, what in zip(got[1:], ref[1:], ("grad", "g", "jac", "cost terms")):
                assert np.array_equal(a, b), (order, what, int(np.argmax(a != b)))
    finally:
        hostemu_lib.set_wave_order(-1)


@pytest.mark.parametrize("terrain", ["planar", "stairs"])
def test_hessian_early_run_is_what_the_recorder_saw(model, terrain):
    """hipnlp_eval_hess through host buffers stores the entries [0, early_run) of a knot's block behind barrier number early_phase of the
    Hessian program (hipnlp.hip, HArgs::early_run).  The run is RECORDED (HessLayout::pos_phase: every entry is emitted exactly once — a
    second emission is refused — so the phase of its emission is when it is final), not declared: every entry of the run is emitted in an
    earlier phase than the barrier, the run is maximal, it covers the point columns (the first 120 variables of a knot, first in
    (column, row) order) and at least three of the six phases are still to run behind it."""
    from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings
    from hostemu_lib import HostEmu
    e = HostEmu((stairs_settings if terrain == "stairs" else periodic_step_settings)(4, model), model)
    ph, run, barrier, (rows, cols) = e.hess_phases()
    assert ph.size == (1878 if terrain == "stairs" else 1539) and ph.max() == 5 and (ph != 255).all()
    assert 1 <= barrier <= 3 and (ph[:run] < barrier).all() and ph[run] >= barrier
    assert run == (732 if terrain == "stairs" else 394) and barrier == (3 if terrain == "stairs" else 2)
    assert (cols < 120).sum() <= run and (cols[:run] < 130).all() and (cols[run:] >= 120).all()     # the point columns (and a few base columns behind them) are the run
    assert run <= 3 * 256       # (three positions per thread of the 256-thread kernel)
