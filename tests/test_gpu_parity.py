"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.
Tolerance: fp64, max |a-b| / max(1,|b|) <= 1e-11 for every entry of f, grad f, g, jac g."""
import os

import numpy as np
import pytest

from hippopt_amd import _abi
from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings, stairs_settings
from hippopt_amd.synthetic import make_workload, place_on_step_flanks
from diag_lib import diag_overrides

pytestmark = pytest.mark.gpu
TOL = 1e-11


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


@pytest.fixture(scope="module")
def HipNlp():
    from hippopt_amd.hipnlp import HipNlp as cls
    return cls


@pytest.mark.parametrize("maker", [periodic_step_settings, single_step_settings])
@pytest.mark.parametrize("horizon", [2, 3, 7, 30])
def test_callback_matches_oracle(model, HipNlp, maker, horizon):
    from oracle_lib import Oracle
    st = maker(horizon, model)
    x, p = make_workload(st, model, batch=1, seed=1000 + horizon)
    eng = HipNlp(st, model)
    orc = Oracle(st, model)
    assert (eng.n, eng.m, eng.nnz, eng.np) == (orc.n, orc.m, orc.nnz, orc.np)
    ir, jc = eng.sparsity()
    iro, jco = orc.sparsity()
    assert np.array_equal(ir, iro) and np.array_equal(jc, jco)
    eng.set_params(p)
    f, grad, g, jac = eng.eval(x)
    fo, grado, go, jaco = orc.eval(x[0], p[0])
    assert rel(f[0], fo) < TOL
    assert rel(grad[0], grado) < TOL
    assert rel(g[0], go) < TOL
    assert rel(jac[0], jaco) < TOL
    lbx, ubx, lbg, ubg = eng.bounds()
    lbo, ubo = orc.bounds(p[0])
    assert np.array_equal(lbg, lbo) and np.array_equal(ubg, ubo)
    names, terms = eng.cost_terms()
    assert np.allclose(terms[0], orc.cost_terms(), rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize("horizon", [3, 50])
def test_smooth_terrain_matches_oracle(model, HipNlp, horizon):
    """Stairs configuration (main_walking_on_stairs.py, BASELINE configs[4]: smooth two-step terrain, N = 50): contact points on
    the flanks of the bumps (batch row 0) and anywhere (row 1); an oriented, offset, differently-sharp step in the mix."""
    from oracle_lib import Oracle
    st = stairs_settings(horizon, model)
    if horizon == 3:
        st.terrain_steps[1].update(orientation=-0.7, position=(0.8, 0.2, 0.03), edge_sharpness=3, side_sharpness=4)
    x, p = make_workload(st, model, batch=2, seed=2000 + horizon)
    place_on_step_flanks(x[0], st, seed=horizon)
    eng, orc = HipNlp(st, model, batch=2), Oracle(st, model)
    assert (eng.n, eng.m, eng.nnz, eng.np) == (orc.n, orc.m, orc.nnz, orc.np)
    ir, jc = eng.sparsity()
    iro, jco = orc.sparsity()
    assert np.array_equal(ir, iro) and np.array_equal(jc, jco)
    eng.set_params(p)
    f, grad, g, jac = eng.eval(x)
    for b in range(2):
        fo, grado, go, jaco = orc.eval(x[b], p[b])
        assert rel(f[b], fo) < TOL and rel(grad[b], grado) < TOL and rel(g[b], go) < TOL and rel(jac[b], jaco) < TOL
    assert np.abs(jac[0]).max() > 100.0   # the flank derivatives really are exercised (planar entries are O(1))


def test_minimize_modes_and_intended_joint_cost(model, HipNlp):
    from oracle_lib import Oracle
    st = periodic_step_settings(5, model)
    st.final_state_expression_type = _abi.EXPR_MINIMIZE
    st.final_state_expression_weight = 3.0
    st.periodicity_expression_type = _abi.EXPR_MINIMIZE
    st.periodicity_expression_weight = 0.7
    st.joint_reg_as_coded = False
    st.contacts_centroid_cost_multiplier = 100.0
    x, p = make_workload(st, model, batch=1, seed=77)
    eng, orc = HipNlp(st, model), Oracle(st, model)
    eng.set_params(p)
    f, grad, g, jac = eng.eval(x)
    fo, grado, go, jaco = orc.eval(x[0], p[0])
    assert (eng.m, eng.nnz) == (orc.m, orc.nnz)
    assert rel(f[0], fo) < TOL and rel(grad[0], grado) < TOL and rel(g[0], go) < TOL and rel(jac[0], jaco) < TOL


def test_random_configurations_both_kernel_variants(model, HipNlp):
    """A seeded sweep over what selects code paths — terrain, horizon (incl. N = 2, 3: no interior knot / one), batch, the three modes of
    the horizon-end expressions, the J6 reading, the kernel variant (HIPNLP_WAVES) and the reduction (in the launch / separate kernel)
    — every result entrywise against the oracle.  The same x / p go through both variants: equal to 1e-13, f bitwise per variant
    pair of reductions."""
    import os
    from oracle_lib import Oracle
    # (HIPNLP_SWEEP_SEED / HIPNLP_SWEEP_CASES: longer one-off sweeps on a GPU box; the suite runs ten cases of one seed)
    rng = np.random.RandomState(int(os.environ.get("HIPNLP_SWEEP_SEED", "20260")))
    modes = (_abi.EXPR_SKIP, _abi.EXPR_SUBJECT_TO, _abi.EXPR_MINIMIZE)
    for case in range(int(os.environ.get("HIPNLP_SWEEP_CASES", "10"))):
        stairs = bool(rng.randint(2))
        N = int(rng.choice([2, 3, 4, 7, 11]))
        st = (stairs_settings if stairs else periodic_step_settings)(N, model)
        st.final_state_expression_type = modes[rng.randint(3)]
        st.final_state_expression_weight = float(rng.uniform(0.5, 3.0))
        st.periodicity_expression_type = modes[rng.randint(3)]
        st.periodicity_expression_weight = float(rng.uniform(0.5, 3.0))
        st.joint_reg_as_coded = bool(rng.randint(2))
        B = int(rng.choice([1, 2, 3]))
        x, p = make_workload(st, model, batch=B, seed=300 + case)
        if stairs:
            place_on_step_flanks(x[:1], st, seed=case)
        orc = Oracle(st, model)
        refs = [orc.eval(x[b], p[b]) for b in range(B)]
        outs = {}
        for waves, sep in ((4, 0), (4, 1), (8, 0)):
            with diag_overrides(HIPNLP_WAVES=waves, HIPNLP_SEPARATE_REDUCE=sep) as lib:   # (kernel variants by force: the diagnostic build)
                eng = HipNlp(st, model, batch=B, library=lib)
            assert (eng.n, eng.m, eng.nnz) == (orc.n, orc.m, orc.nnz), case
            eng.set_params(p)
            f, grad, g, jac = eng.eval(x)
            for b in range(B):
                fo, grado, go, jaco = refs[b]
                assert rel(f[b], fo) < TOL and rel(grad[b], grado) < TOL and rel(g[b], go) < TOL and rel(jac[b], jaco) < TOL, (case, waves, sep, b)
            outs[(waves, sep)] = (f.copy(), grad.copy(), g.copy(), jac.copy())
        # exact Hessian of the Lagrangian at the same points, random multipliers / objective factors (entrywise against the oracle)
        from hess_util import hess_mismatch, triplets_to_dict
        lam, sig = rng.standard_normal((B, orc.m)), rng.uniform(0.0, 2.0, B)
        hr, hc = eng.hess_sparsity()
        hv = eng.eval_hess(x, sig, lam)
        for b in range(B):
            err, where = hess_mismatch(triplets_to_dict(hr, hc, hv[b]), triplets_to_dict(*orc.hess(x[b], p[b], float(sig[b]), lam[b])), diag_scaled=stairs)
            assert err <= (1e-9 if stairs else TOL), (case, b, where)
        assert np.array_equal(outs[(4, 0)][0], outs[(4, 1)][0])     # the two reductions of one kernel: the same tree
        assert all(np.array_equal(a, b_) for a, b_ in zip(outs[(4, 0)][1:], outs[(4, 1)][1:]))
        for a, b_ in zip(outs[(4, 0)], outs[(8, 0)]):
            assert rel(a, b_) < 1e-13, case


def test_batch_and_determinism(model, HipNlp):
    from oracle_lib import Oracle
    st = periodic_step_settings(12, model)
    x, p = make_workload(st, model, batch=5, seed=5)
    p[3, :] = make_workload(st, model, batch=1, seed=99)[1][0]  # one trajectory with different parameters
    eng, orc = HipNlp(st, model, batch=5), Oracle(st, model)
    eng.set_params(p)
    out1 = eng.eval(x)
    out2 = eng.eval(x)
    for a, b in zip(out1, out2):
        assert np.array_equal(a, b)  # bitwise repeatable
    for b in range(5):
        fo, grado, go, jaco = orc.eval(x[b], p[b])
        assert rel(out1[0][b], fo) < TOL and rel(out1[1][b], grado) < TOL
        assert rel(out1[2][b], go) < TOL and rel(out1[3][b], jaco) < TOL


def test_new_x_cache_and_errors(model, HipNlp):
    from hippopt_amd.hipnlp import HipNlpError
    st = single_step_settings(4, model)
    x, p = make_workload(st, model, batch=1, seed=3)
    eng = HipNlp(st, model)
    with pytest.raises(HipNlpError) as e:
        eng.eval(x)
    assert e.value.code == _abi.E_PARAMS  # parameters must be set before solve (opti_solver.py:447-450)
    eng.set_params(p)
    f1, *_ = eng.eval(x)
    f2, *_ = eng.eval(x * 0.0, new_x=False)  # cached
    assert f1[0] == f2[0]
    bad = x.copy()
    bad[0, _abi.NXK + 130:_abi.NXK + 134] = 0.0  # zero quaternion -> normalisation divides by zero
    with pytest.raises(HipNlpError) as e:
        eng.eval(bad)
    assert e.value.code == _abi.E_NUMERIC


def test_config5_stairs_200x16_matches_oracle(model, HipNlp):
    """BASELINE config 5 at its own shape: walking on stairs (smooth two-step terrain), N = 200 knots x 16 batched initial guesses
    (base trajectory + N(0, 0.02^2) per guess) in ONE launch.  Entrywise against the oracle for four of the sixteen
    trajectories (the oracle takes ~0.1 s per trajectory at this horizon), pattern and bounds for the problem, and every
    trajectory of the batch against the same trajectory evaluated alone and in other batch positions (no cross-talk inside the batch)."""
    from oracle_lib import Oracle
    N, B = 200, 16
    st = stairs_settings(N, model)
    base_x, base_p = make_workload(st, model, batch=1, seed=1005)
    x = np.repeat(base_x, B, axis=0)
    for b in range(B):
        x[b] += 0.02 * np.random.RandomState(2000 + b).standard_normal(x.shape[1])
    place_on_step_flanks(x[:2], st, seed=5)   # two guesses with contact points on the flanks of the bumps
    p = np.repeat(base_p, B, axis=0)
    eng, orc = HipNlp(st, model, batch=B), Oracle(st, model)
    assert (eng.n, eng.m, eng.nnz, eng.np) == (orc.n, orc.m, orc.nnz, orc.np) == (189 * N + 6, orc.m, orc.nnz, 79 * N + 326)
    ir, jc = eng.sparsity()
    iro, jco = orc.sparsity()
    assert np.array_equal(ir, iro) and np.array_equal(jc, jco)
    eng.set_params(p)
    f, grad, g, jac = eng.eval(x)
    lbx, ubx, lbg, ubg = eng.bounds()
    lbo, ubo = orc.bounds(p[0])
    assert np.array_equal(lbg, lbo) and np.array_equal(ubg, ubo)
    for b in (0, 1, 7, 15):
        fo, grado, go, jaco = orc.eval(x[b], p[b])
        assert rel(f[b], fo) < TOL and rel(grad[b], grado) < TOL and rel(g[b], go) < TOL and rel(jac[b], jaco) < TOL, b
    assert np.abs(jac[0]).max() > 100.0   # the flank derivatives are exercised
    # (the single trajectory runs on the eight-wave kernel, the batch on the four-wave one: two instantiations of the same source,
    #  equal to the last bits but not bitwise — the compiler contracts a handful of terrain entries differently)
    one = HipNlp(st, model, batch=1)
    one.set_params(p[:1])
    for b in (3, 12):
        f1, grad1, g1, jac1 = one.eval(x[b:b + 1])
        assert rel(f1[0], f[b]) < 1e-13 and rel(grad1[0], grad[b]) < 1e-13 and rel(g1[0], g[b]) < 1e-13 and rel(jac1[0], jac[b]) < 1e-13
    # the same trajectory anywhere in the batch gives bitwise the same values (same kernel)
    xs = np.repeat(x[5:6], B, axis=0)
    fs, grads, gs_, jacs = eng.eval(xs)
    for b in (1, 9, 15):
        assert fs[b] == fs[0] and np.array_equal(jacs[b], jacs[0]) and np.array_equal(gs_[b], gs_[0]) and np.array_equal(grads[b], grads[0])
    assert np.array_equal(jacs[0], jac[5]) and fs[0] == f[5]


def test_config4_periodic_100_matches_oracle(model, HipNlp):
    """BASELINE config 4 at its own shape (the bench workload: periodic walking, N = 100, seed 1004): every entry of
    f, grad f, g, jac g against the oracle, pattern and bounds included."""
    from oracle_lib import Oracle
    st = periodic_step_settings(100, model)
    x, p = make_workload(st, model, batch=1, seed=1004)
    eng, orc = HipNlp(st, model), Oracle(st, model)
    assert (eng.n, eng.m, eng.nnz) == (orc.n, orc.m, orc.nnz) == (18906, 27505, 137879)
    ir, jc = eng.sparsity()
    iro, jco = orc.sparsity()
    assert np.array_equal(ir, iro) and np.array_equal(jc, jco)
    eng.set_params(p)
    f, grad, g, jac = eng.eval(x)
    fo, grado, go, jaco = orc.eval(x[0], p[0])
    assert rel(f[0], fo) < TOL and rel(grad[0], grado) < TOL and rel(g[0], go) < TOL and rel(jac[0], jaco) < TOL
    lbx, ubx, lbg, ubg = eng.bounds()
    lbo, ubo = orc.bounds(p[0])
    assert np.array_equal(lbg, lbo) and np.array_equal(ubg, ubo)
    names, terms = eng.cost_terms()
    assert np.allclose(terms[0], orc.cost_terms(), rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize("vary_first", [False, True])
def test_config4_x64_full_occupancy_has_no_cross_talk(model, HipNlp, vary_first):
    """BASELINE config 4 as the throughput legs launch it: N = 100 x 64 trajectories, 6 400 knot workgroups of the four-wave kernels, four
    (CCS) / five (varying-first) per CU over several rounds, the reducer workgroups of the trajectories among them.  Size-independent
    properties: one trajectory entrywise against the oracle, every trajectory bit for bit what it is in another order of the batch (nothing
    depends on the position in the launch or on the workgroups that share a CU), every output finite."""
    from oracle_lib import Oracle
    st = periodic_step_settings(100, model)
    B = 64
    x, p = make_workload(st, model, batch=B, seed=1004)
    eng, orc = HipNlp(st, model, batch=B, jac_varying_first=vary_first), Oracle(st, model)
    eng.set_params(p)
    f, grad, g, jac = (np.array(a) for a in eng.eval(x))
    assert all(np.isfinite(a).all() for a in (f, grad, g, jac))
    ir, jc = eng.sparsity()
    iro, jco = orc.sparsity()
    b = 37
    fo, grado, go, jaco = orc.eval(x[b], p[b])
    order = np.lexsort((ir, jc)) if vary_first else np.arange(ir.size)   # (the varying-first handle reports its own entry order)
    assert rel(f[b], fo) < TOL and rel(grad[b], grado) < TOL and rel(g[b], go) < TOL
    assert np.array_equal(ir[order], iro) and np.array_equal(jc[order], jco) and rel(jac[b][order], jaco) < TOL
    perm = np.random.RandomState(7).permutation(B)
    eng.set_params(p[perm])
    f2, grad2, g2, jac2 = eng.eval(x[perm])
    assert np.array_equal(f2, f[perm]) and np.array_equal(grad2, grad[perm]) and np.array_equal(g2, g[perm]) and np.array_equal(jac2, jac[perm])


def test_large_horizon_properties(model, HipNlp):
    """N = 100 (BASELINE config 4): size-independent properties — defect linearity in dt, zero defects on a
    trapezoid-consistent trajectory, Jacobian consistent with finite differences of g along a random direction."""
    st = periodic_step_settings(100, model)
    x, p = make_workload(st, model, batch=1, seed=1004)
    eng = HipNlp(st, model)
    eng.set_params(p)
    f, grad, g, jac = eng.eval(x)
    ir, jc = eng.sparsity()
    rng = np.random.RandomState(0)
    d = rng.standard_normal(eng.n)
    eps = 1e-6
    _, _, gp, _ = eng.eval(x + eps * d, want=("g",))
    _, _, gm, _ = eng.eval(x - eps * d, want=("g",))
    jd = np.zeros(eng.m)
    np.add.at(jd, ir, jac[0] * d[jc])
    fd = (gp[0] - gm[0]) / (2 * eps)
    assert np.max(np.abs(fd - jd)) / max(1.0, np.max(np.abs(fd))) < 1e-6
    fp, *_ = eng.eval(x + eps * d, want=("f",))
    fm, *_ = eng.eval(x - eps * d, want=("f",))
    assert abs((fp[0] - fm[0]) / (2 * eps) - grad[0] @ d) / max(1.0, abs(grad[0] @ d)) < 1e-6
    # base-position defect rows vanish on a trapezoid roll-out of the base position
    xr = x[0].copy()
    dt = st.time_step
    K = _abi.NXK
    for k in range(1, 100):
        xr[K * k + 127:K * k + 130] = xr[K * (k - 1) + 127:K * (k - 1) + 130] + 0.5 * dt * (xr[K * (k - 1) + 120:K * (k - 1) + 123] + xr[K * k + 120:K * k + 123])
    _, _, gr, _ = eng.eval(xr[None, :], want=("g",))
    blocks = {b[0]: b for b in eng.row_blocks()}
    name, first, rows, k0, nk = blocks["base_position_dynamics"]
    assert np.max(np.abs(gr[0][first:first + rows * nk])) < 1e-13


@pytest.mark.parametrize("name", ["planner_periodic_N3", "planner_single_N3", "planner_costends_N2", "planner_stairs_N3", "planner_ramp_N3"])
def test_gpu_matches_reference_planner_fixtures(model, HipNlp, name):
    """HIP path against the golden vectors generated by executing the reference's planner code on the CasADi-API stand-in
    (tools/gen_planner_fixtures.py): same x, p -> g (reference row order), bounds, f, grad f, jac g."""
    import json
    import os
    from test_golden_planner import GOLD, settings_for
    z = np.load(os.path.join(GOLD, name + ".npz"))
    st = settings_for(json.loads(str(z["meta"])), model)
    eng = HipNlp(st, model)
    eng.set_params(z["p"][None, :])
    f, grad, g, jac = eng.eval(z["x"][None, :])
    assert rel(g[0], z["g"]) < TOL and rel(grad[0], z["grad"]) < TOL and rel(f[0], float(z["f"])) < TOL
    lbx, ubx, lbg, ubg = eng.bounds()
    assert np.array_equal(lbg, z["lbg"]) and np.array_equal(ubg, z["ubg"])
    ir, jc = eng.sparsity()
    J = {(int(r), int(c)): v for r, c, v in zip(ir, jc, jac[0])}
    for r, c, v in zip(z["jac_row"], z["jac_col"], z["jac_val"]):
        got = J.pop((int(r), int(c)), None)
        if got is None:
            assert abs(v) < 1e-12
        else:
            assert abs(got - v) <= TOL * max(1.0, abs(v))
    assert max((abs(v) for v in J.values()), default=0.0) < 1e-12
    if "hess_dirs" in z.files:   # Hessian-vector products of the reference planner's graph
        from test_golden_planner import hessian_times
        hr, hc = eng.hess_sparsity()
        hv = eng.eval_hess(z["x"][None, :], float(z["hess_sigma"]), z["hess_lambda"][None, :])[0]
        assert rel(hessian_times(hr, hc, hv, eng.n, z["hess_dirs"]), z["hess_times_dirs"]) < (1e-9 if ("stairs" in name or "ramp" in name) else TOL)


def test_planner_solve_plumbing(model):
    """Planner -> HipNlpSolver -> engine callbacks -> NLP driver (SciPy trust-constr stand-in when IPOPT is absent) -> Output.
    A few iterations from a synthetic guess must run end to end, keep the structure and reduce the constraint violation."""
    from hippopt_amd.kinodyn_settings import single_step_settings
    from hippopt_amd.turnkey_planners.humanoid_kinodynamic import Planner, Settings
    st = Settings.from_numeric(single_step_settings(3, model), solver_options={"max_iter": 15})
    pl = Planner(st, model, error_on_fail=False)   # a handful of iterations: keep the last iterate of the unconverged run
    x, p = make_workload(st, model, batch=1, seed=8)
    guess = pl.get_initial_guess()
    names = pl.optimization_solver._var_index
    guess.from_dict({n: x[0][off:off + size].reshape(shape) for n, (off, size, shape) in names.items()})
    pars = pl.optimization_solver._par_index
    guess.from_dict({n: p[0][off:off + size].reshape(shape) for n, (off, size, shape) in pars.items()})
    pl.optimization_solver.set_initial_guess(guess)      # already mass-normalised synthetic data
    eng = pl.optimization_solver.engine()
    eng.set_params(p)
    _, _, g0, _ = eng.eval(x)
    _, _, lbg, ubg = eng.bounds()
    viol0 = np.max(np.maximum(0, np.maximum(lbg - g0[0], g0[0] - ubg)))
    out = pl.solve()
    assert len(out.values.system) == 3 and np.asarray(out.values.system[0].com).size == 3
    assert set(out.cost_values) == set(eng.cost_terms()[0])
    assert "joint_position_dynamics" in out.constraint_multipliers
    assert out.constraint_multipliers["joint_position_dynamics"].shape == (2, 23)
    xs = np.concatenate([np.asarray(v, float).reshape(-1) for n, v in out.values.to_dict().items() if n in names])
    assert np.isfinite(out.cost_value)
    info = pl.optimization_solver._last_info
    assert info["constr_violation"] < viol0
    # detect_simple_bounds (main_periodic_step.py:109-110): the driver saw the reduced problem, the output still names every constraint
    assert eng.lifted and info["nlp"]["simple_bounds_lifted"] == 70 * 2 + 47 + 81 == eng.m_full - eng.m and info["nlp"]["m"] == eng.m
    assert out.constraint_multipliers["joint_velocity_bounds"].shape == (3, 23) and info["callbacks"]["evaluations"] > 0
    full = Planner(st, model, error_on_fail=False)
    full.optimization_solver._detect_simple_bounds = False
    assert full.optimization_solver.nlp_view() is full.optimization_solver.engine()


@pytest.mark.parametrize("maker,horizon,batch", [(periodic_step_settings, 100, 1), (periodic_step_settings, 9, 70), (single_step_settings, 2, 3),
                                                  (stairs_settings, 12, 2), (stairs_settings, 7, 90)])
def test_detect_simple_bounds_handle_is_the_reduced_problem(model, HipNlp, maker, horizon, batch):
    """HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS: the handle evaluates the NLP nlpsol hands to IPOPT under {"detect_simple_bounds": True}
    (main_periodic_step.py:109-110).  Against the full handle on the same inputs, both kernel variants and both terrains: g, jac g,
    the pattern and the row bounds are the full problem's with the single-variable rows taken out (bitwise), the variable bounds are
    those rows' bounds, f and grad f are unchanged, and the exact Hessian takes the reduced multipliers."""
    st = maker(horizon, model)
    x, p = make_workload(st, model, batch=batch, seed=4100 + horizon)
    full, red = HipNlp(st, model, batch=batch), HipNlp(st, model, batch=batch, detect_simple_bounds=True)
    full.set_params(p)
    red.set_params(p)
    simple, var = full.simple_rows()
    kept, lb_full, ub_full = red.lift_map()
    keep_rows = np.nonzero(simple == 0)[0]
    assert red.lifted and not full.lifted and red.m_full == full.m == full.m_full and red.n_lifted == int(simple.sum())
    assert np.array_equal(np.nonzero(kept >= 0)[0], keep_rows) and np.array_equal(kept[keep_rows], np.arange(red.m))
    ir, jc = full.sparsity()
    irr, jcr = red.sparsity()
    keep_entries = np.nonzero(simple[ir] == 0)[0]
    assert np.array_equal(keep_rows[irr], ir[keep_entries]) and np.array_equal(jcr, jc[keep_entries])
    lbx, ubx, lbg, ubg = red.bounds()
    _, _, lbg_f, ubg_f = full.bounds()
    assert np.array_equal(lbg, lbg_f[keep_rows]) and np.array_equal(ubg, ubg_f[keep_rows]) and np.array_equal(lb_full, lbg_f) and np.array_equal(ub_full, ubg_f)
    lifted = np.nonzero(simple)[0]
    rl, ru = np.full(full.n, -np.inf), np.full(full.n, np.inf)
    np.maximum.at(rl, var[lifted], lbg_f[lifted])
    np.minimum.at(ru, var[lifted], ubg_f[lifted])
    assert np.array_equal(lbx, rl) and np.array_equal(ubx, ru)
    f, grad, g, jac = red.eval(x)
    ff, gradf, gf, jacf = full.eval(x)
    assert np.array_equal(f, ff) and np.array_equal(grad, gradf)
    assert np.array_equal(g, gf[:, keep_rows]) and np.array_equal(jac, jacf[:, keep_entries])
    # a trial-point call (f + g only) and a cached Jacobian request on the reduced handle
    x2 = x + 1e-3
    f2, _, g2, _ = red.eval(x2, want=("f", "g"))
    _, _, _, j2 = red.eval(x2, new_x=False, want=("jac",))
    ff2, _, gf2, jf2 = full.eval(x2)
    assert np.array_equal(f2, ff2) and np.array_equal(g2, gf2[:, keep_rows]) and np.array_equal(j2, jf2[:, keep_entries])
    # exact Hessian: the lifted rows are linear, their multipliers do not exist in the reduced problem
    lam = np.random.RandomState(5).standard_normal((batch, red.m))
    lam_full = np.zeros((batch, full.m))
    lam_full[:, keep_rows] = lam
    hr, hc = red.hess_sparsity()
    hrf, hcf = full.hess_sparsity()
    assert np.array_equal(hr, hrf) and np.array_equal(hc, hcf)
    assert np.array_equal(red.eval_hess(x, 0.7, lam), full.eval_hess(x, 0.7, lam_full))


def test_in_launch_reduction_stress(model, HipNlp):
    """Cost reduction (per-knot partials -> f, per-term costs) over many back-to-back launches at a batch that fills the chip
    several times, alternating between two iterates: bitwise identical results every time (no stale partials, fixed tree)."""
    st = periodic_step_settings(40, model)
    batch = 96
    x, p = make_workload(st, model, batch=batch, seed=17)
    eng = HipNlp(st, model, batch=batch)
    eng.set_params(p)
    f0, *_ = eng.eval(x, want=("f",))
    _, t0 = eng.cost_terms()
    assert np.all(np.isfinite(f0)) and np.allclose(t0.sum(axis=1), f0, rtol=1e-13)
    x2 = x + 1e-3 * np.random.RandomState(3).standard_normal(x.shape)   # alternate iterates: a stale record would show
    f2, *_ = eng.eval(x2, want=("f",))
    assert not np.array_equal(f2, f0)
    for i in range(150):
        f, *_ = eng.eval(x if i % 2 == 0 else x2, want=("f",))
        assert np.array_equal(f, f0 if i % 2 == 0 else f2)
    eng.eval(x, want=("f",))
    _, t1 = eng.cost_terms()
    assert np.array_equal(t0, t1)
    # the reducer workgroup inside the launch and the separate reduction kernel (kept for long trajectories / launches) sum in the
    # same fixed tree: bitwise the same f and per-term costs
    assert eng.kernels_per_eval() == 1
    with diag_overrides(HIPNLP_SEPARATE_REDUCE=1) as lib:
        sep = HipNlp(st, model, batch=batch, library=lib)
    assert sep.kernels_per_eval() == 2
    sep.set_params(p)
    fsep, *_ = sep.eval(x, want=("f",))
    _, tsep = sep.cost_terms()
    assert np.array_equal(fsep, f0) and np.array_equal(tsep, t0)
    from oracle_lib import Oracle
    orc = Oracle(st, model)
    for b in (0, 37, 95):
        fo, _ = orc.eval_fg(x[b], p[b])
        assert abs(f0[b] - fo) / max(1.0, abs(fo)) < 1e-11


def test_best_iterate_callback_fallback(model):
    """use_opti_callback (planner.py:56-63, opti_solver.py:451-520): a run that stops at the iteration limit counts as a failure
    (CasADi's Opti raises); without a criterion the solver plugin raises, with BestCost & AcceptablePrimalInfeasibility it returns
    the best iterate the callback saved, with its cost, per-term costs and multipliers."""
    from hippopt_amd.hipnlp_solver import HipFailure
    from hippopt_amd.kinodyn_settings import single_step_settings
    from hippopt_amd.turnkey_planners.humanoid_kinodynamic import Planner, Settings

    def planner(**kw):
        st = Settings.from_numeric(single_step_settings(3, model), solver_options={"max_iter": 12}, **kw)
        pl = Planner(st, model)
        x, p = make_workload(st, model, batch=1, seed=8)
        guess = pl.get_initial_guess()
        guess.from_dict({n: x[0][off:off + size].reshape(shape) for n, (off, size, shape) in pl.optimization_solver._var_index.items()})
        guess.from_dict({n: p[0][off:off + size].reshape(shape) for n, (off, size, shape) in pl.optimization_solver._par_index.items()})
        pl.optimization_solver.set_initial_guess(guess)
        return pl
    with pytest.raises(HipFailure):
        planner().solve()
    with pytest.raises(HipFailure):     # a criterion nothing satisfies: "the callback did not manage to save ..."
        planner(use_opti_callback=True, acceptable_constraint_violation=0.0).solve()
    pl = planner(use_opti_callback=True, acceptable_constraint_violation=np.inf)
    out = pl.solve()
    cb = pl.optimization_solver._callback
    assert cb.best_iteration is not None and np.isclose(out.cost_value, cb.best_cost)
    assert set(out.cost_values) == set(pl.optimization_solver.engine().cost_terms()[0])
    assert abs(sum(out.cost_values.values()) - out.cost_value) <= 1e-9 * max(1.0, abs(out.cost_value))
    assert out.constraint_multipliers["joint_position_dynamics"].shape == (2, 23)
    # the values are the saved iterate (forces / momenta are multiplied back by the mass on the way out: compare the joints)
    assert np.allclose(np.asarray(out.values.system[1].kinematics.joints.positions).reshape(-1), cb.best_x[189 + 157:189 + 180])


def test_maximum_sizes_and_shard_consistency(model, HipNlp):
    """Large shapes: N = 800 (an 8 m walk at the stairs configuration's knot density, BASELINE config 5 x 4) and a batch that fills
    the chip many times.  Checked through size-independent properties: (i) every knot of a long horizon equals the same knot
    evaluated in a short horizon with the same neighbours (rows are knot-local), (ii) knot shards tile the full result,
    (iii) batch entries are independent of their batch position, (iv) sampled entries against the oracle."""
    from oracle_lib import Oracle
    N = 800
    st = single_step_settings(N, model)
    x, p = make_workload(st, model, batch=1, seed=31)
    eng = HipNlp(st, model)
    eng.set_params(p)
    f, grad, g, jac = eng.eval(x)
    assert np.all(np.isfinite(g)) and np.all(np.isfinite(jac)) and np.isfinite(f[0])
    assert eng.n == 189 * N + 6 and eng.nnz == eng.dims.nnz_knot * (N - 2) + (eng.nnz - eng.dims.nnz_knot * (N - 2))
    # (ii) two shards reproduce the full jac / grad runs bit for bit
    import torch
    xd = torch.from_numpy(x[0]).cuda()
    pieces_j, pieces_g = [], []
    for kb, ke in ((0, 300), (300, 800)):
        sh = HipNlp(st, model, knot_begin=kb, knot_end=ke)
        sh.set_params(p)
        d = sh.dims
        gs = torch.zeros(int(d.shard_grad), dtype=torch.float64, device="cuda")
        js = torch.zeros(int(d.shard_nnz), dtype=torch.float64, device="cuda")
        stage = torch.zeros((ke - kb) * 550, dtype=torch.float64, device="cuda")
        fs = torch.zeros(1, dtype=torch.float64, device="cuda")
        sh.eval_device_shard(xd.data_ptr(), fs.data_ptr(), gs.data_ptr(), stage.data_ptr(), js.data_ptr())
        torch.cuda.synchronize()
        pieces_j.append(js.cpu().numpy())
        pieces_g.append(gs.cpu().numpy())
    assert np.array_equal(np.concatenate(pieces_j), jac[0]) and np.array_equal(np.concatenate(pieces_g), grad[0])
    # (iv) oracle on the full problem is too slow at N = 800 for every entry; it still finishes once
    orc = Oracle(st, model)
    fo, grado, go, jaco = orc.eval(x[0], p[0])
    assert rel(f[0], fo) < TOL and rel(g[0], go) < TOL and rel(jac[0], jaco) < TOL and rel(grad[0], grado) < TOL
    # (iii) a large batch: entry b of a batch equals the same trajectory evaluated alone
    st2 = periodic_step_settings(20, model)
    B = 2048
    xs, ps = make_workload(st2, model, batch=8, seed=5)
    xb, pb = np.tile(xs, (B // 8, 1)), np.tile(ps, (B // 8, 1))
    big = HipNlp(st2, model, batch=B)
    big.set_params(pb)
    fb, gradb, gb, jacb = big.eval(xb)
    one = HipNlp(st2, model, batch=8)
    one.set_params(ps)
    f1, grad1, g1, jac1 = one.eval(xs)
    for b in (0, 9, 1027, B - 1):
        assert np.array_equal(jacb[b], jac1[b % 8]) and np.array_equal(gb[b], g1[b % 8]) and fb[b] == f1[b % 8]


@pytest.mark.parametrize("lifted,compact", [(False, False), (True, False), (False, True), (True, True)])
def test_sharded_callback_reassembly_on_gpu(model, HipNlp, lifted, compact):
    """(lifted: the knot shards of the detect_simple_bounds problem — what a sharded solve of the reference's scripts evaluates.)
    ShardedCallback on the GPU at world size 1: the fused shard buffer of the whole horizon goes through the
    library's one-launch reassembly (hipnlp_reassemble) and must equal the unsharded callback bit for bit; f is the in-kernel sum.
    x CHANGES ON EVERY CALL and there are no warm-up repeats: a result that is not ordered behind its own shard evaluation (the
    evaluation on one stream, the reassembly on another) would show the previous iterate's values.
    compact: the exchange without the constants of jac g (varying-first handles: the fused buffer carries the varying runs, the reassembled
    buffer holds the constants, hipnlp_reassemble_scatter writes around them) — also across a set_params that changes dt."""
    import torch
    from hippopt_amd.sharded import ShardedCallback, hip_constants, hip_shard_backend, hip_shard_info
    N = 24
    st = periodic_step_settings(N, model)
    x, p = make_workload(st, model, batch=1, seed=77)
    full = HipNlp(st, model, detect_simple_bounds=lifted, jac_varying_first=compact)
    full.set_params(p)
    full.set_constant_jacobian(False)            # the reference stores every entry on every call
    dev = torch.device("cuda", 0)
    sh = HipNlp(st, model, knot_begin=0, knot_end=N, detect_simple_bounds=lifted, jac_varying_first=compact)
    sh.set_params(p)
    assert (sh.m < sh.m_full) == lifted
    info = hip_shard_info(sh, 0, N, compact)
    cb = ShardedCallback(N, sh.n, sh.m, sh.nnz, info, hip_shard_backend(sh, compact), dev, **(hip_constants(sh) if compact else {}))
    if compact:
        lay = sh.jac_vary_layout()
        assert info["jvary"] == lay["shard_len"] == lay["total"] == int((~sh.jac_constant_mask()).sum()) and lay["shard_off"] == 0
        assert cb.shard_len == 1 + info["glen"] + info["jvary"] + info["slen"] and info["jvary"] < 0.62 * info["jlen"] and info["slen"] == sh.m + 0 * N
        assert (info["stage_rows"] >= 0).all() and np.array_equal(np.sort(info["stage_rows"]), np.arange(sh.m))   # every row of g once, no padding
    rng = np.random.RandomState(4)
    xs = [x[0] + 1e-2 * i * rng.standard_normal(x.shape[1]) for i in range(6)]
    xds = [torch.from_numpy(xi).to(dev) for xi in xs]
    torch.cuda.synchronize()
    p2 = p.copy()
    p2[:, 24 * N + 3 + 105 + 105] *= 1.3             # dt: the constants of the trapezoid defects change with it
    for params in ((p, p2) if compact else (p,)):
        if params is p2:
            full.set_params(p2)
            sh.set_params(p2)     # (no refresh_constants(): the callback sees the engine's parameter generation move and refreshes by itself)
        got = []
        for xd in xds:   # back to back, results copied out on the caller's stream without an explicit synchronisation in between
            fs, grads, jacs, gs = cb(xd)
            got.append((fs.clone(), grads.clone(), jacs.clone(), gs.clone()))
        torch.cuda.synchronize()
        for xi, (fs, grads, jacs, gs) in zip(xs, got):
            f, grad, g, jac = full.eval(xi[None, :])
            assert float(fs) == f[0]
            assert np.array_equal(grads.cpu().numpy(), grad[0]) and np.array_equal(jacs.cpu().numpy(), jac[0]) and np.array_equal(gs.cpu().numpy(), g[0])


def test_launch_number_wrap(model, HipNlp):
    """The launch number tags the published cost partials and is the generation of the non-finite flags; at 2^31 - 1 it starts
    over (flags and tags cleared).  HIPNLP_DEBUG_SEQ0 starts the count just below: evaluations across the wrap give bitwise the
    values of a fresh handle, and a non-finite x is still reported on both sides of it."""
    import os
    from hippopt_amd.hipnlp import HipNlpError
    st = periodic_step_settings(8, model)
    x, p = make_workload(st, model, batch=2, seed=5)
    x2 = x + 1e-2 * np.random.RandomState(7).standard_normal(x.shape)
    ref = HipNlp(st, model, batch=2)
    ref.set_params(p)
    r1, r2 = [a.copy() for a in ref.eval(x)], [a.copy() for a in ref.eval(x2)]
    with diag_overrides(HIPNLP_DEBUG_SEQ0=2 ** 31 - 4) as lib:
        eng = HipNlp(st, model, batch=2, library=lib)
    eng.set_params(p)
    bad = x.copy()
    bad[1, 40] = np.nan
    for i in range(8):   # launches 2^31 - 3 ... then 1, 2, ...
        xi, r = (x, r1) if i % 2 == 0 else (x2, r2)
        got = eng.eval(xi)
        assert all(np.array_equal(a, b_) for a, b_ in zip(got, r)), i
        if i in (1, 5):
            with pytest.raises(HipNlpError):
                eng.eval(bad)
    hv_ref = ref.eval_hess(x, 1.0, np.ones((2, ref.m)))
    for _ in range(6):
        assert np.array_equal(eng.eval_hess(x, 1.0, np.ones((2, eng.m))), hv_ref)


def test_hessian_values_straight_into_a_registered_array(model, HipNlp):
    """hipnlp_eval_hess with the caller's value array inside a range registered with hipnlp_host_register: the kernel stores into it
    directly (no pinned block, no host copy); the same values as the plain path, also after the array has been unregistered again."""
    st = periodic_step_settings(12, model)
    x, p = make_workload(st, model, batch=2, seed=64)
    eng = HipNlp(st, model, batch=2)
    eng.set_params(p)
    lam = np.random.RandomState(3).standard_normal((2, eng.m))
    ref = eng.eval_hess(x, [0.7, 1.3], lam)
    out = np.full_like(ref, np.nan)
    eng.register_outputs([out])
    try:
        got = eng.eval_hess(x, [0.7, 1.3], lam, out=out)
        assert got is out and np.array_equal(out, ref)
        x2 = x + 1e-3
        assert np.array_equal(eng.eval_hess(x2, [0.7, 1.3], lam, out=out), eng.eval_hess(x2, [0.7, 1.3], lam))
    finally:
        eng.unregister_outputs([out])
    out[:] = np.nan
    assert np.array_equal(eng.eval_hess(x, [0.7, 1.3], lam, out=out), ref)


@pytest.mark.parametrize("early_grad", [False, True])
def test_early_outputs_into_registered_arrays(model, HipNlp, early_grad):
    """hipnlp_set_early_outputs: the new-x call (eval_f) already fills the registered g / jac arrays the later eval_g / eval_jac_g
    calls pass; alternating iterates, so values left over from the previous x would show; a cached call with ANOTHER array is
    still served correctly (by a fresh evaluation), and unregistering turns the mode off for that array.
    grad f is NOT touched early unless asked for (on = 2): IPOPT's adapter hands eval_grad_f the storage of its own gradient vector
    of the current iterate, which must survive the evaluation of trial points."""
    st = periodic_step_settings(9, model)
    x, p = make_workload(st, model, batch=1, seed=43)
    x2 = x + 1e-2 * np.random.RandomState(2).standard_normal(x.shape)
    eng = HipNlp(st, model)
    eng.set_params(p)
    ref = {0: [a.copy() for a in eng.eval(x)], 1: [a.copy() for a in eng.eval(x2)]}
    outs = [np.zeros_like(a) for a in ref[0]]
    eng.register_outputs(outs)
    try:
        eng.set_early_outputs(True, grad=early_grad)
        f_, grad_, g_, jac_ = outs
        for i in range(6):
            xi, r = (x, ref[0]) if i % 2 == 0 else (x2, ref[1])
            prev = ref[1] if i % 2 == 0 else ref[0]
            eng.eval(xi, new_x=True, want=("f",), out=(f_, None, None, None))
            if i > 0:   # the arrays have been seen: g and jac already hold this x's values, before being asked for
                assert np.array_equal(g_, r[2]) and np.array_equal(jac_, r[3])
                assert np.array_equal(grad_, r[1] if early_grad else prev[1])     # the caller's gradient of the PREVIOUS point is intact
            eng.eval(xi, new_x=False, want=("g",), out=(None, None, g_, None))
            eng.eval(xi, new_x=False, want=("grad",), out=(None, grad_, None, None))
            eng.eval(xi, new_x=False, want=("jac",), out=(None, None, None, jac_))
            assert f_[0] == r[0][0] and np.array_equal(g_, r[2]) and np.array_equal(grad_, r[1]) and np.array_equal(jac_, r[3])
        other = np.zeros_like(jac_)   # a cached request with an array the library has not filled
        eng.eval(x2, new_x=True, want=("f",), out=(f_, None, None, None))
        eng.eval(x2, new_x=False, want=("jac",), out=(None, None, None, other))
        assert np.array_equal(other, ref[1][3])
    finally:
        eng.set_early_outputs(False)
        eng.unregister_outputs(outs)
    got = eng.eval(x, new_x=True)   # back to the plain path, unregistered arrays
    assert np.array_equal(got[3], ref[0][3])


@pytest.mark.parametrize("compact", [False, True])
def test_peer_exchange_equals_the_gathered_callback(model, HipNlp, compact):
    """PeerExchange (peer stores into IPC-shared output buffers + flags, no collective, no reassembly pass) at world size 1: the
    rank pushes into its own buffer; bitwise the unsharded callback, on alternating iterates and both buffer parities.
    compact: the varying runs of jac g only (varying-first handles; hipnlp_eval_device_peers_vary for the folded push), the receiver's
    two buffers hold the constants — refreshed after a set_params that changes dt."""
    import torch
    from hippopt_amd.sharded import PeerExchange, ShardedCallback, hip_constants, hip_shard_backend, hip_shard_info
    N = 24
    st = periodic_step_settings(N, model)
    x, p = make_workload(st, model, batch=1, seed=78)
    full = HipNlp(st, model, jac_varying_first=compact)
    full.set_params(p)
    full.set_constant_jacobian(False)
    dev = torch.device("cuda", 0)
    sh = HipNlp(st, model, knot_begin=0, knot_end=N, jac_varying_first=compact)
    sh.set_params(p)
    cb = ShardedCallback(N, sh.n, sh.m, sh.nnz, hip_shard_info(sh, 0, N, compact), hip_shard_backend(sh, compact), dev, **(hip_constants(sh) if compact else {}))
    rng = np.random.RandomState(5)
    xs = [x[0] + 1e-2 * i * rng.standard_normal(x.shape[1]) for i in range(5)]
    # push kernel behind the shard evaluation / stores folded into the evaluation (hipnlp_eval_device_peers); every rank receives /
    # gather_to_root (rank 0 alone receives, back-flags pace the others: here rank 0 is the only rank)
    for engine, root_only in ((None, False), (sh, False), (None, True), (sh, True)):
        px = PeerExchange(cb, engine=engine, root_only=root_only)     # (set-up ends with the one-word handshake)
        got = []
        for xi in xs:
            fs, grads, jacs, gs = px(torch.from_numpy(xi).to(dev))
            got.append((fs.clone(), grads.clone(), jacs.clone(), gs.clone()))
        assert not px.timed_out()
        px.check()
        for xi, (fs, grads, jacs, gs) in zip(xs, got):
            f, grad, g, jac = full.eval(xi[None, :])
            assert float(fs) == f[0]
            assert np.array_equal(grads.cpu().numpy(), grad[0]) and np.array_equal(jacs.cpu().numpy(), jac[0]) and np.array_equal(gs.cpu().numpy(), g[0])
        assert px.bytes_sent_per_step() == 0       # one rank: nothing leaves the device
        if compact:     # a parameter change: new constants into the receiver's buffers, then the same comparison under the new parameters
            p2 = p.copy()
            p2[:, 24 * N + 3 + 105 + 105] *= 0.8
            for e in (full, sh):
                e.set_params(p2)  # (no px.refresh_constants(): keyed on the engine's parameter generation, as the callback's own buffer)
            for xi in xs[:3]:
                fs, grads, jacs, gs = px(torch.from_numpy(xi).to(dev))
                torch.cuda.synchronize()
                f, grad, g, jac = full.eval(xi[None, :])
                assert float(fs) == f[0] and np.array_equal(jacs.cpu().numpy(), jac[0]) and np.array_equal(gs.cpu().numpy(), g[0])
            for e in (full, sh):
                e.set_params(p)
            # (what a rank would send: a third less than the exchange of every entry)
            me = cb.infos[0]
            assert px.max_bytes_sent_per_step() == 0 and me["jvary"] < 0.62 * me["jlen"]
        px.close()
    # the collective's gather_to_root at world size 1 (the parameters went back to p above: refreshed by the call itself)
    fs, grads, jacs, gs = cb.to_root(torch.from_numpy(xs[1]).to(dev))
    torch.cuda.synchronize()
    f, grad, g, jac = full.eval(xs[1][None, :])
    assert float(fs) == f[0] and np.array_equal(jacs.cpu().numpy(), jac[0]) and np.array_equal(gs.cpu().numpy(), g[0])


def test_a_wait_that_gives_up_is_sticky_and_poisons_the_step(model, HipNlp):
    """hipnlp_peer_wait on a flag that never comes: the status word is raised and STAYS raised over later, successful waits; grad / jac /
    g of the step are NaN throughout, not only f (a timeout must not look like an ordinary evaluation)."""
    import ctypes as C
    import torch
    lib = HipNlp.__module__ and __import__("hippopt_amd.hipnlp", fromlist=["load_library"]).load_library()
    vp = C.c_void_p
    lib.hipnlp_peer_wait.argtypes = [vp, C.c_int, C.c_ulonglong, vp, C.c_int64, vp, vp]
    dev = torch.device("cuda", 0)
    world, tot = 2, 1000
    flags = torch.zeros(64, dtype=torch.int64, device=dev)
    flags[0] = 5                                   # rank 0 signalled step 5, rank 1 never does
    out = torch.ones(tot + world + 1, dtype=torch.float64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream()
    assert lib.hipnlp_peer_wait(flags.data_ptr(), world, 5, out.data_ptr(), tot, status.data_ptr(), stream.cuda_stream) == 0
    stream.synchronize()
    assert int(status.item()) == 1 and bool(torch.isnan(out[:tot]).all()) and bool(torch.isnan(out[tot + world]))
    flags[1] = 6
    flags[0] = 6
    out.fill_(2.0)
    assert lib.hipnlp_peer_wait(flags.data_ptr(), world, 6, out.data_ptr(), tot, status.data_ptr(), stream.cuda_stream) == 0
    stream.synchronize()
    assert int(status.item()) == 1                 # sticky
    # ... and it travels: a rank whose status word is raised signals with the poison bit (hipnlp_peer_signal_checked), the wait on the
    # receiving side passes at once and poisons ITS step (a push into a buffer that may still have been read must not pass for an evaluation)
    vp = C.c_void_p
    lib.hipnlp_peer_signal_checked.argtypes = [vp, C.c_int, C.c_int, C.c_ulonglong, vp, vp]
    flags2 = torch.zeros(world, dtype=torch.int64, device=dev)
    ftab = torch.tensor([flags2.data_ptr()], dtype=torch.int64, device=dev)
    status2 = torch.zeros(1, dtype=torch.int32, device=dev)
    for r in range(world):   # rank 1 signals with a raised status word, the others with a clean one
        assert lib.hipnlp_peer_signal_checked(ftab.data_ptr(), 1, r, 9, (status if r == 1 else status2).data_ptr(), stream.cuda_stream) == 0
    out2 = torch.zeros(tot + world + 1, dtype=torch.float64, device=dev)
    assert lib.hipnlp_peer_wait(flags2.data_ptr(), world, 9, out2.data_ptr(), tot, status2.data_ptr(), stream.cuda_stream) == 0
    stream.synchronize()
    assert int(status2.item()) == 1 and bool(torch.isnan(out2[:tot]).all()) and bool(torch.isnan(out2[tot + world]))
    assert int(flags2[1].item()) & 0xFFFFFFFF == 9 and int(flags2[0].item()) == 9
    assert float(out[tot + world]) == 4.0 and bool((out[:tot] == 2.0).all())


@pytest.mark.parametrize("vary", [False, True])
@pytest.mark.parametrize("terrain,waves,split", [("planar", 8, 12), ("planar", 4, 9), ("stairs", 8, 15), ("stairs", 4, 1)])
def test_evaluation_that_stores_into_every_ranks_buffer(model, HipNlp, terrain, waves, split, vary):
    """hipnlp_eval_device_peers: the knot kernel of a shard handle writes the shard's entries at their FINAL positions into the
    [grad | jac | g | f partials | f] buffer of every rank.  Two shard handles of one process play two ranks (uneven shards, both
    kernel variants, both terrains); both buffers must hold, bit for bit, the unsharded callback's grad / jac / g, and the cost
    partials the all-gather path sums.
    vary: hipnlp_eval_device_peers_vary on varying-first handles — the buffers hold the constant entries of jac g
    (hipnlp_fill_jac_constants, whole horizon, by a SHARD handle), the kernels store the varying runs at their places in the pattern."""
    import torch
    N = 24
    st = (stairs_settings if terrain == "stairs" else periodic_step_settings)(N, model)
    x, p = make_workload(st, model, batch=1, seed=91)
    if terrain == "stairs":
        place_on_step_flanks(x, st, seed=91)
    dev = torch.device("cuda", 0)
    with diag_overrides(HIPNLP_WAVES=waves) as lib:
        full = HipNlp(st, model, library=lib, jac_varying_first=vary)
        shards = [HipNlp(st, model, knot_begin=0, knot_end=split, library=lib, jac_varying_first=vary),
                  HipNlp(st, model, knot_begin=split, knot_end=N, library=lib, jac_varying_first=vary)]
    for e in [full] + shards:
        e.set_params(p)
    full.set_constant_jacobian(False)
    n, m, nnz = full.n, full.m, full.nnz
    tot, world = n + nnz + m, 2
    bufs = [torch.full((tot + world + 1,), float("nan"), dtype=torch.float64, device=dev) for _ in range(world)]

    def constants_in_place():
        if vary:   # (one buffer by each shard handle: either knows the whole pattern)
            for r, b in enumerate(bufs):
                shards[r].fill_jac_constants(b.data_ptr() + 8 * n, True, stream.cuda_stream)
    table = torch.tensor([b.data_ptr() for b in bufs], dtype=torch.int64, device=dev)
    flags = torch.zeros(64, dtype=torch.int64, device=dev)
    ftab = torch.tensor([flags.data_ptr()], dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream()
    lib = full.lib
    import ctypes as C
    vp = C.c_void_p
    lib.hipnlp_peer_signal.argtypes = [vp, C.c_int, C.c_int, C.c_ulonglong, vp]
    lib.hipnlp_peer_wait.argtypes = [vp, C.c_int, C.c_ulonglong, vp, C.c_int64, vp, vp]
    rng = np.random.RandomState(3)
    for step in (1, 2, 3):
        xi = x[0] + 1e-2 * step * rng.standard_normal(x.shape[1])
        xd = torch.from_numpy(xi).to(dev)
        with torch.cuda.stream(stream):
            constants_in_place()
            for r, e in enumerate(shards):
                (e.eval_device_peers_vary if vary else e.eval_device_peers)(xd.data_ptr(), table.data_ptr(), world, r, stream=stream.cuda_stream)
                # (one process: both "ranks" signal into the one flag array)
                assert lib.hipnlp_peer_signal(ftab.data_ptr(), 1, 0, step, stream.cuda_stream) == 0
        stream.synchronize()
        flags[1] = flags[0]   # the wait below is that of "rank 0": both slots must carry the step
        with torch.cuda.stream(stream):
            assert lib.hipnlp_peer_wait(flags.data_ptr(), world, step, bufs[0].data_ptr(), tot, status.data_ptr(), stream.cuda_stream) == 0
        stream.synchronize()
        assert int(status.item()) == 0
        f, grad, g, jac = full.eval(xi[None, :])
        for b in bufs:
            o = b.cpu().numpy()
            assert np.array_equal(o[:n], grad[0]) and np.array_equal(o[n:n + nnz], jac[0]) and np.array_equal(o[n + nnz:tot], g[0])
            assert np.isfinite(o[tot:tot + world]).all() and abs((o[tot] + o[tot + 1]) - f[0]) <= 1e-12 * abs(f[0])
        assert float(bufs[0][tot + world]) == float(bufs[0][tot]) + float(bufs[0][tot + 1])   # hipnlp_peer_wait: partials in rank order
        for b in bufs:
            b.fill_(float("nan"))


def test_peer_mode_random_shardings(model, HipNlp):
    """hipnlp_eval_device_peers over a seeded sweep: horizon, 2-5 "ranks" (shard handles of one process) with random cut points — shards of
    one knot, the first and the last knot alone or together with interior ones —, both terrains, both kernel variants, the three
    modes of the horizon-end expressions: every rank's buffer bitwise the unsharded callback's grad / jac / g, the shard costs
    summing to f."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(int(os.environ.get("HIPNLP_SWEEP_SEED", "77")))
    modes = (_abi.EXPR_SKIP, _abi.EXPR_SUBJECT_TO, _abi.EXPR_MINIMIZE)
    stream = torch.cuda.Stream()
    for case in range(int(os.environ.get("HIPNLP_SWEEP_CASES", "6"))):
        stairs = bool(rng.randint(2))
        N = int(rng.choice([2, 3, 5, 8, 13]))
        world = int(rng.randint(2, min(N, 5) + 1))
        cuts = [0] + sorted(rng.choice(np.arange(1, N), size=world - 1, replace=False).tolist()) + [N]
        st = (stairs_settings if stairs else periodic_step_settings)(N, model)
        st.final_state_expression_type, st.periodicity_expression_type = modes[rng.randint(3)], modes[rng.randint(3)]
        x, p = make_workload(st, model, batch=1, seed=600 + case)
        waves = int(rng.choice([4, 8]))
        lift = bool(rng.randint(2))     # the detect_simple_bounds layout: shards of the reduced problem
        with diag_overrides(HIPNLP_WAVES=waves) as lib:
            full = HipNlp(st, model, detect_simple_bounds=lift, library=lib)
            shards = [HipNlp(st, model, knot_begin=cuts[r], knot_end=cuts[r + 1], detect_simple_bounds=lift, library=lib) for r in range(world)]
        for e in [full] + shards:
            e.set_params(p)
        n, m, nnz = full.n, full.m, full.nnz
        tot = n + nnz + m
        bufs = [torch.full((tot + world + 1,), float("nan"), dtype=torch.float64, device=dev) for _ in range(world)]
        table = torch.tensor([b.data_ptr() for b in bufs], dtype=torch.int64, device=dev)
        xd = torch.from_numpy(x[0]).to(dev)
        with torch.cuda.stream(stream):
            for r, e in enumerate(shards):
                e.eval_device_peers(xd.data_ptr(), table.data_ptr(), world, r, stream=stream.cuda_stream)
        stream.synchronize()
        f, grad, g, jac = full.eval(x)
        for b in bufs:
            o = b.cpu().numpy()
            assert np.array_equal(o[:n], grad[0]) and np.array_equal(o[n:n + nnz], jac[0]) and np.array_equal(o[n + nnz:tot], g[0]), (case, cuts)
            assert np.isfinite(o[tot:tot + world]).all() and abs(o[tot:tot + world].sum() - f[0]) <= 1e-12 * max(1.0, abs(f[0])), (case, cuts)


def test_two_ranks_on_one_gpu_rehearsal():
    """`bench.py --gpus 2` as the driver starts it, with both ranks on device 0 and gloo as the rendezvous (BENCH_REHEARSAL=1: RCCL
    refuses two ranks on one device).  Not a measurement: it runs the N > 1 code paths across two PROCESSES — self-spawned ranks,
    the knot-sharded all-gather path as `value`, the peer exchange (IPC handles opened by the other process, flags, both buffer
    parities: checked bit for bit against the all-gather result inside bench.py) and the shared host sink."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--no-cpu-baseline",
                          "--no-hessian", "--no-host"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    # the LAST line is the short one the driver parses; the line before it (BENCH_DETAILS ...) carries every leg in full
    lines = out.stdout.strip().splitlines()
    final = json.loads(lines[-1])
    assert len(lines[-1]) < 6000 and final["n_gpus"] == 2 and final["value"] > 0 and final["config"]["exchange"] == "all_gather"
    assert set(final["exchanges"]) >= {"all_gather", "peer_store", "peer_direct", "gather_to_root", "host_sink"} and "REHEARSAL" in final["config"]["backend"]
    detail_lines = [ln for ln in lines if ln.startswith("BENCH_DETAILS ")]
    assert len(detail_lines) == 1
    line = json.loads(detail_lines[0][len("BENCH_DETAILS "):])
    assert line["n_gpus"] == 2 and line["value"] > 0 and abs(line["value"] - final["value"]) <= 1e-6 * line["value"]
    side = {k: line.get(k) or line["config"].get(k) for k in ("peer_store", "peer_direct", "gather_to_root", "host_sink", "shard_resident")}
    for k in ("peer_store", "peer_direct", "gather_to_root"):
        assert side[k] and "error" not in side[k], (k, side[k])
        assert side[k]["verified"].startswith("bitwise") and not side[k]["timed_out"]
        assert side[k]["efficiency_vs_n_independent_gpus"] is None            # ranks share a device: an efficiency would mean nothing
        assert side[k]["bytes_sent_per_rank_per_step"] > 0
    # (the maximum over the ranks: a gather_to_root's rank 0 sends nothing, the other rank sends its shard once)
    assert 500_000 < side["gather_to_root"]["bytes_sent_per_rank_per_step"] < side["peer_store"]["bytes_sent_per_rank_per_step"] * 1.01
    assert side["peer_store"]["bytes_sent_per_rank_per_step"] > 1_000_000
    assert side["host_sink"] and "error" not in side["host_sink"], side["host_sink"]
    # a rehearsal (ranks sharing a device) never promotes a peer exchange: `value` is the all-gather path, and the line says why
    legs = {"all_gather": line["all_gather"], "peer_store": side["peer_store"], "peer_direct": side["peer_direct"]}
    assert all(v["steps"] == line["steps"] == 30 for v in legs.values())
    assert line["config"]["exchange"] == "all_gather" and line["config"]["peer_paths_eligible_for_value"].startswith("no")
    assert abs(line["ms_per_step"] - legs["all_gather"]["ms_per_step"]) < 1e-9
    assert line["all_gather"]["rccl_ranks"] == 2 and line["all_gather"]["efficiency_vs_n_independent_gpus"] is None
    # BASELINE's two multi-GPU configurations are legs of the same line: config 4 = ONE periodic trajectory of 100 knots cut over the
    # ranks (strong scaling), config 5 = the stairs, 200 knots x 16 guesses dealt over the ranks with the outputs on rank 0
    c4, c5 = line["config4_strong"], line["config5"]
    assert "error" not in c4 and c4["ranks"] == 2
    for k in ("all_gather", "gather_to_root", "peer_direct"):
        assert c4[k] and "error" not in c4[k] and c4[k]["knots_per_s"] > 0 and c4[k]["efficiency_vs_one_gpu"] is None, (k, c4[k])
    assert "error" not in c5 and c5["one_gpu_whole_job"]["knots_per_s"] > 0 and c5["local_only"]["knots_per_s"] > 0
    for k in ("rccl_gather", "peer_direct_to_root"):
        assert "error" not in c5[k] and c5[k]["knots_per_s"] > 0 and c5[k]["verified"].startswith("every trajectory"), (k, c5[k])
        assert c5[k]["efficiency_vs_one_gpu"] is None and c5[k]["bytes_sent_per_rank_per_step"] > 8 * 8 * 37806
    assert not c5["peer_direct_to_root"]["timed_out"]


def test_host_path_want_mask_lazy_fetch_and_views(model, HipNlp):
    """hipnlp_eval / hipnlp_eval_pinned: the kernel stores the wanted outputs straight into the pinned block; the others stay in
    HBM and are fetched when a later new_x = 0 call asks.  Every combination must return exactly the values of one full evaluation
    at the SAME x (alternating iterates, so a stale block would show), whatever the prefetch set."""
    from hippopt_amd.hipnlp import HipNlpError
    st = periodic_step_settings(9, model)
    x, p = make_workload(st, model, batch=2, seed=41)
    x2 = x + 1e-2 * np.random.RandomState(1).standard_normal(x.shape)
    eng = HipNlp(st, model, batch=2)
    eng.set_params(p)
    ref = {0: [a.copy() for a in eng.eval(x)], 1: [a.copy() for a in eng.eval(x2)]}
    names = ("f", "grad", "g", "jac")
    for prefetch in (("f", "grad", "g"), (), ("f", "grad", "g", "jac"), ("jac",)):
        eng.set_prefetch(prefetch)
        for i, first in enumerate(names):
            xi, r = (x, ref[0]) if i % 2 == 0 else (x2, ref[1])
            out = eng.eval(xi, want=(first,))                       # a new evaluation that asks for ONE output
            assert [o is not None for o in out] == [n == first for n in names]
            assert np.array_equal(out[i], r[i])
            for j, later in enumerate(names):                        # the others afterwards, from the cached evaluation
                o2 = eng.eval(xi * 0.0, new_x=False, want=(later,))  # (x is ignored when new_x = 0)
                assert np.array_equal(o2[j], r[j]), (prefetch, first, later)
        views = eng.eval_pinned(x2, want=("g", "jac"))
        assert views[0] is None and views[1] is None and np.array_equal(views[2], ref[1][2]) and np.array_equal(views[3], ref[1][3])
        vf = eng.eval_pinned(x2, new_x=False, want=("f", "grad"))
        assert np.array_equal(vf[0], ref[1][0]) and np.array_equal(vf[1], ref[1][1])
    names_t, terms = eng.cost_terms()
    assert np.allclose(terms.sum(axis=1), ref[1][0], rtol=1e-13)
    # a non-finite evaluation: the outputs are filled BEFORE the error code comes back (IPOPT looks at the NaNs and cuts the step)
    bad = x.copy()
    bad[1, 130:134] = 0.0   # zero quaternion in trajectory 1
    with pytest.raises(HipNlpError) as e:
        eng.eval(bad)
    assert e.value.code == _abi.E_NUMERIC
    f, grad, g, jac = eng.eval(bad, nan_ok=True)
    assert not np.all(np.isfinite(jac[1])) and np.array_equal(jac[0], ref[0][3][0]) and np.array_equal(g[0], ref[0][2][0])
    f, grad, g, jac = eng.eval(x)   # and the handle recovers
    assert np.array_equal(jac, ref[0][3]) and np.array_equal(f, ref[0][0])
    # caller arrays registered with the library are stored to by the kernel itself (no staging copy): same values, also on a
    # cached (new_x = 0) request for an output the evaluation has already delivered into the caller's array
    import ctypes as C
    reg = [np.zeros_like(a) for a in ref[0]]
    for a in reg[1:]:
        assert eng.lib.hipnlp_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), None) == 0
    try:
        for xi, r in ((x2, ref[1]), (x, ref[0])):
            out = eng.eval(xi, out=tuple(reg))
            for o, want in zip(out, r):
                assert np.array_equal(o, want)
            reg[3][:] = 0.0
            eng.eval(xi, new_x=False, want=("jac",), out=(None, None, None, reg[3]))
            assert np.array_equal(reg[3], r[3])
    finally:
        for a in reg[1:]:
            eng.lib.hipnlp_host_unregister(C.c_void_p(a.ctypes.data))


@pytest.mark.parametrize("order", ["ccs", "ccs, constants in place", "varying first"])
def test_host_sink_shards_store_into_one_registered_buffer(model, HipNlp, order):
    """HostSink (SURVEY §5's alternative to the all-gather): the knot kernels of two shard handles store their g / jac / grad f
    straight into ONE registered host buffer in the reference's order — no collective, no staging copy.  Bitwise equal to the
    unsharded evaluation; the rank partial costs sum to f.
    Both orders of a knot block; with the constants in place (the default of varying-first handles; opt-in in CCS order) every shard
    handle fills the constants of ITS knots into the sink at its first sight of it — and again after a set_params — and stores the
    varying entries from then on; a destination in host memory is never read back by the launch (no check over PCIe)."""
    import torch
    from hippopt_amd.sharded import HostSink, knot_range
    N = 17
    vf = order == "varying first"
    st = periodic_step_settings(N, model)
    x, p = make_workload(st, model, batch=1, seed=52)
    full = HipNlp(st, model, jac_varying_first=vf)
    full.set_constant_jacobian(False)
    sink = HostSink("hipnlp_test_sink_%d" % __import__("os").getpid(), full.n, full.m, full.nnz, world=2, rank=0)
    xs = [x[0] + 1e-2 * i * np.random.RandomState(8).standard_normal(x.shape[1]) for i in range(3)]
    xd = [torch.from_numpy(xi).cuda() for xi in xs]
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    shards = []
    for r in range(2):
        kb, ke = knot_range(N, 2, r)
        sh = HipNlp(st, model, knot_begin=kb, knot_end=ke, jac_varying_first=vf)
        if order != "ccs":
            sh.set_constant_jacobian(True)
        shards.append(sh)
    p2 = p.copy()
    p2[:, 24 * N + 3 + 105 + 105] *= 1.25             # dt
    fills = 0
    for params in (p, p2):
        full.set_params(params)
        for sh in shards:
            sh.set_params(params)
        for i in range(3):
            f, grad, g, jac = full.eval(xs[i][None, :])
            fp, gradp, gp, jacp = sink.pointers()
            for r, sh in enumerate(shards):
                sh.eval_device(xd[i].data_ptr(), sink.dev + 8 * r, gradp, gp, jacp, stream=stream.cuda_stream)
                stream.synchronize()
            fparts, grad_h, jac_h, g_h = sink.views()
            assert np.array_equal(grad_h, grad[0]) and np.array_equal(jac_h, jac[0]) and np.array_equal(g_h, g[0]), (order, i)
            assert abs(sink.f() - f[0]) <= 1e-13 * max(1.0, abs(f[0]))
        fills += 1
        for sh in shards:
            stats = sh.host_stats()
            assert stats["constant_fills"] == (0 if order == "ccs" else fills) and stats["constant_slices_healed"] == 0, (order, stats)
    sink.close()


def test_profile_runs_and_kernel_count(model, HipNlp):
    """hipnlp_kernels_per_eval (1 up to 256 knots per trajectory in launches of up to 32768 knots, 2 beyond) and the run-bracketed event timing."""
    import torch
    st = periodic_step_settings(12, model)
    x, p = make_workload(st, model, batch=1, seed=3)
    small = HipNlp(st, model)
    small.set_params(p)
    assert small.kernels_per_eval() == 1
    assert HipNlp(st, model, batch=64).kernels_per_eval() == 1   # (the cost is summed by a reducer workgroup per trajectory)
    assert HipNlp(periodic_step_settings(300, model), model).kernels_per_eval() == 2   # more than 256 knots per trajectory
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x).to(dev)
    outs = [torch.empty(k, dtype=torch.float64, device=dev) for k in (1, small.n, small.m, small.nnz)]
    small.profile_begin_runs(4, 8)
    for _ in range(40):
        small.eval_device(xd.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(),
                          stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    kern_ms, launch_ms, n = small.profile_end()
    assert n == 32 and 0.0 < kern_ms < 1.0 and launch_ms >= kern_ms
    f, grad, g, jac = small.eval(x)
    assert float(outs[0].cpu()[0]) == f[0] and np.array_equal(outs[3].cpu().numpy(), jac[0])


# ---- exact Hessian of the Lagrangian (hipnlp_eval_hess) ----------------------------------------------------------------------------
@pytest.mark.parametrize("maker", [periodic_step_settings, single_step_settings])
@pytest.mark.parametrize("horizon", [2, 3, 9])
def test_hessian_matches_oracle(model, HipNlp, maker, horizon):
    from hess_util import hess_mismatch, triplets_to_dict
    from oracle_lib import Oracle
    st = maker(horizon, model)
    x, p = make_workload(st, model, batch=1, seed=3000 + horizon)
    o = Oracle(st, model)
    eng = HipNlp(st, model)
    eng.set_params(p)
    lam = np.random.RandomState(horizon).standard_normal(o.m)
    ir, jc = eng.hess_sparsity()
    vals = eng.eval_hess(x, 0.8, lam)[0]
    err, where = hess_mismatch(triplets_to_dict(ir, jc, vals), triplets_to_dict(*o.hess(x[0], p[0], 0.8, lam)))
    assert err <= TOL, where
    # the callback quartet is untouched by a Hessian evaluation in between
    f, grad, g, jac = eng.eval(x)
    fo, grado, go, jaco = o.eval(x[0], p[0])
    assert rel(f[0], fo) <= TOL and rel(g[0], go) <= TOL and rel(jac[0], jaco) <= TOL and rel(grad[0], grado) <= TOL


def test_hessian_compact_layout_is_the_full_layout_bit_for_bit(model, HipNlp):
    """The exact-Hessian kernel of the planar terrain has two instantiations: the full scratch (two workgroups per CU; launches of at
    most 512 workgroups) and the compact scratch with the lite tables (three per CU; longer launches).  Same program, same arithmetic:
    the same bits — each forced through HIPNLP_HESS_LAYOUT on a launch the other would get, cost-mode horizon ends included (the
    compact layout reads their tables from global memory) — and the automatic choice on a launch of more than 512 workgroups against
    the oracle."""
    import os
    from hess_util import hess_mismatch, triplets_to_dict
    from oracle_lib import Oracle
    st = periodic_step_settings(40, model)
    st.final_state_expression_type = st.periodicity_expression_type = _abi.EXPR_MINIMIZE
    st.final_state_expression_weight, st.periodicity_expression_weight = 2.0, 0.5
    B = 14                                      # 560 workgroups: compact by itself
    x, p = make_workload(st, model, batch=B, seed=3200)
    rng = np.random.RandomState(6)
    vals = {}
    for layout in ("full", "compact", None):
        if layout:
            with diag_overrides(HIPNLP_HESS_LAYOUT=layout) as lib:
                eng = HipNlp(st, model, batch=B, library=lib)
        else:
            eng = HipNlp(st, model, batch=B)
        eng.set_params(p)
        if "lam" not in vals:
            vals["lam"], vals["sig"] = rng.standard_normal((B, eng.m)), rng.uniform(0.0, 1.5, B)
        vals[layout] = eng.eval_hess(x, vals["sig"], vals["lam"])
        ir, jc = eng.hess_sparsity()
        eng.close()
    assert np.array_equal(vals["full"], vals["compact"]) and np.array_equal(vals[None], vals["compact"])
    o = Oracle(st, model)
    for b in (0, B - 1):
        err, where = hess_mismatch(triplets_to_dict(ir, jc, vals[None][b]), triplets_to_dict(*o.hess(x[b], p[b], vals["sig"][b], vals["lam"][b])))
        assert err <= TOL, (b, where)


def test_hessian_cost_modes_batch_and_shards(model, HipNlp):
    from hess_util import hess_mismatch, triplets_to_dict
    from oracle_lib import Oracle
    st = periodic_step_settings(6, model)
    st.final_state_expression_type = st.periodicity_expression_type = _abi.EXPR_MINIMIZE
    st.final_state_expression_weight, st.periodicity_expression_weight = 2.0, 0.5
    B = 3
    x, p = make_workload(st, model, batch=B, seed=3100)
    o = Oracle(st, model)
    eng = HipNlp(st, model, batch=B)
    eng.set_params(p)
    rng = np.random.RandomState(5)
    lam = rng.standard_normal((B, o.m))
    sig = np.array([1.0, 0.3, 0.0])
    ir, jc = eng.hess_sparsity()
    vals = eng.eval_hess(x, sig, lam)
    assert np.array_equal(vals, eng.eval_hess(x, sig, lam))   # deterministic
    for b in range(B):
        err, where = hess_mismatch(triplets_to_dict(ir, jc, vals[b]), triplets_to_dict(*o.hess(x[b], p[b], sig[b], lam[b])))
        assert err <= TOL, (b, where)
    # knot shards: every shard handle evaluates the blocks of its own knots (bitwise the same values)
    parts_v, parts_r, parts_c = [], [], []
    for kb, ke in ((0, 2), (2, 5), (5, 6)):
        sh = HipNlp(st, model, batch=B, knot_begin=kb, knot_end=ke)
        sh.set_params(p)
        r, c = sh.hess_sparsity()
        parts_r.append(r); parts_c.append(c); parts_v.append(sh.eval_hess(x, sig, lam))
    assert np.array_equal(np.concatenate(parts_r), ir) and np.array_equal(np.concatenate(parts_c), jc)
    assert np.array_equal(np.concatenate(parts_v, axis=1), vals)


def test_hessian_errors_and_device_pointers(model, HipNlp):
    import torch
    from hippopt_amd.hipnlp import HipNlpError
    st = periodic_step_settings(4, model)
    x, p = make_workload(st, model, batch=2, seed=3200)
    eng = HipNlp(st, model, batch=2)
    with pytest.raises(HipNlpError):
        eng.eval_hess(x, 1.0, np.zeros((2, eng.m)))   # parameters not set
    eng.set_params(p)
    lam = np.random.RandomState(2).standard_normal((2, eng.m))
    host = eng.eval_hess(x, 1.0, lam)
    xd, ld, sd = torch.tensor(x, device="cuda"), torch.tensor(lam, device="cuda"), torch.ones(2, dtype=torch.float64, device="cuda")
    out = torch.zeros((2, eng.hess_nnz()), dtype=torch.float64, device="cuda")
    eng.eval_hess_device(xd.data_ptr(), sd.data_ptr(), ld.data_ptr(), out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), host)
    xb = x.copy()
    xb[1, 130:134] = 0.0   # zero quaternion: the normalisation divides by zero
    with pytest.raises(HipNlpError) as e:
        eng.eval_hess(xb, 1.0, lam)
    assert e.value.code == -5


@pytest.mark.parametrize("oriented", [False, True])
def test_hessian_smooth_terrain_matches_oracle(model, HipNlp, oriented):
    """Stairs configuration (BASELINE config 5's terrain): tolerance 1e-9 entrywise, as for the smooth-terrain Jacobian."""
    from hess_util import hess_mismatch, triplets_to_dict
    from oracle_lib import Oracle
    st = stairs_settings(4, model)
    if oriented:
        st.terrain_steps[0]["orientation"] = 0.4
        st.terrain_steps[1]["orientation"] = -1.1
        st.terrain_steps[1]["position"] = (0.8, 0.2, 0.03)
        st.terrain_steps[1]["edge_sharpness"], st.terrain_steps[1]["side_sharpness"] = 3, 4
    B = 2
    x, p = make_workload(st, model, batch=B, seed=3300)
    place_on_step_flanks(x[:1], st, seed=4)   # trajectory 0 on the flanks of the bumps, trajectory 1 as generated
    o = Oracle(st, model)
    eng = HipNlp(st, model, batch=B)
    eng.set_params(p)
    lam = np.random.RandomState(9).standard_normal((B, o.m))
    ir, jc = eng.hess_sparsity()
    vals = eng.eval_hess(x, 0.8, lam)
    assert np.array_equal(vals, eng.eval_hess(x, 0.8, lam))
    for b in range(B):
        err, where = hess_mismatch(triplets_to_dict(ir, jc, vals[b]), triplets_to_dict(*o.hess(x[b], p[b], 0.8, lam[b])), diag_scaled=True)
        assert err <= 1e-9, (b, where)


@pytest.mark.parametrize("workload", ["periodic", "stairs"])
def test_hessian_at_full_size_is_the_derivative_of_the_lagrangian_gradient(model, HipNlp, workload):
    """BASELINE sizes (N=100 x batch 4; stairs: N=200 x 2), where the oracle is too slow: H d against central differences of the
    engine's OWN Lagrangian gradient  sigma grad f + J^T lambda  (size-independent property; fp64 differences: 1e-6 relative
    to the largest entry of H d; looser on the stairs, whose terrain exponents inflate the truncation error)."""
    from test_golden_planner import hessian_times
    N, B = (100, 4) if workload == "periodic" else (200, 2)
    st = periodic_step_settings(N, model) if workload == "periodic" else stairs_settings(N, model)
    x, p = make_workload(st, model, batch=B, seed=3400)
    eng = HipNlp(st, model, batch=B)
    eng.set_params(p)
    rng = np.random.RandomState(11)
    lam = rng.standard_normal((B, eng.m))
    sigma = 0.6
    ir, jc = eng.sparsity()
    hr, hc = eng.hess_sparsity()
    assert hr.size == eng.hess_nnz() and np.all(hr >= hc) and np.all(hr // 189 == hc // 189)   # block diagonal by knot

    def lag_grad(xx):
        _, grad, _, jac = eng.eval(xx)
        out = sigma * grad
        for b in range(B):
            np.add.at(out[b], jc, jac[b] * lam[b][ir])
        return out
    vals = eng.eval_hess(x, sigma, lam)
    d = rng.standard_normal(x.shape)
    eps = 1e-6
    fd = (lag_grad(x + eps * d) - lag_grad(x - eps * d)) / (2 * eps)
    for b in range(B):
        Hd = hessian_times(hr, hc, vals[b], eng.n, d[b])
        assert np.max(np.abs(Hd - fd[b])) <= (1e-6 if workload == "periodic" else 1e-4) * max(1.0, np.max(np.abs(Hd)))


def test_engine_from_the_reference_objects_fixture(model, HipNlp):
    """tests/golden/from_reference_periodic_N4.npz = what hippopt_amd.from_reference produced from the reference's own Settings /
    Variables objects (tools/gen_from_reference_fixture.py): the engine created from the stored hipnlp_desc BYTES, fed the stored
    p and x, against the oracle."""
    import os
    from oracle_lib import Oracle
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "from_reference_periodic_N4.npz"))
    eng = HipNlp.from_desc(z["desc"].tobytes())
    eng.set_params(z["p"][None, :])
    f, grad, g, jac = eng.eval(z["x"][None, :])
    st = periodic_step_settings(int(z["horizon"]), model)
    orc = Oracle(st, model)
    fo, grado, go, jaco = orc.eval(z["x"], z["p"])
    assert (eng.n, eng.m, eng.nnz) == (orc.n, orc.m, orc.nnz)
    assert rel(f[0], fo) < TOL and rel(grad[0], grado) < TOL and rel(g[0], go) < TOL and rel(jac[0], jaco) < TOL


@pytest.mark.parametrize("seed", [0, 3])
def test_joint_numbering_that_does_not_follow_the_tree(model, HipNlp, seed):
    """A joints_name_list need not list a joint after its parent joint (ergoCub's names torso_pitch before torso_roll): a randomly
    renumbered robot, callback quartet and exact Hessian against the oracle, planar and smooth terrain."""
    from hess_util import hess_mismatch, triplets_to_dict
    from oracle_lib import Oracle
    from test_kernel_body_hostemu import renumbered
    m2 = renumbered(model, np.random.RandomState(seed).permutation(23))
    for maker, N, tol in ((periodic_step_settings, 9, TOL), (stairs_settings, 5, 1e-9)):
        st = maker(N, m2)
        x, p = make_workload(st, m2, batch=2, seed=950 + seed)
        eng, orc = HipNlp(st, m2, batch=2), Oracle(st, m2)
        ir, jc = eng.sparsity()
        iro, jco = orc.sparsity()
        assert np.array_equal(ir, iro) and np.array_equal(jc, jco)
        eng.set_params(p)
        f, grad, g, jac = eng.eval(x)
        lam = np.random.RandomState(seed).standard_normal((2, orc.m))
        hr, hc = eng.hess_sparsity()
        hv = eng.eval_hess(x, 0.7, lam)
        for b in range(2):
            fo, grado, go, jaco = orc.eval(x[b], p[b])
            assert rel(f[b], fo) < TOL and rel(grad[b], grado) < TOL and rel(g[b], go) < TOL and rel(jac[b], jaco) < TOL
            err, where = hess_mismatch(triplets_to_dict(hr, hc, hv[b]), triplets_to_dict(*orc.hess(x[b], p[b], 0.7, lam[b])), diag_scaled=maker is stairs_settings)
            assert err <= tol, (b, where)


def test_sixteen_batched_guesses_through_the_engine(model, HipNlp):
    """SURVEY 8f rank 2 as the row is written: 16 contact-phase descriptions -> ONE [16][n] block of decision vectors
    (hippopt_amd.robot_planning.batched_guess) -> one launch of HipNlp(batch=16) on the stairs configuration (BASELINE config 5's
    terrain, the reference's own horizon N = 50, main_walking_on_stairs.py:70).  Every guess against the same guess evaluated
    alone, three of them against the oracle."""
    from oracle_lib import Oracle
    from hippopt_amd.robot_planning.batched_guess import batched_guess_block
    from test_batched_guess import make_case
    N, B = 50, 16
    st = stairs_settings(N, model)
    rng = np.random.RandomState(50)
    cases = [make_case(g, N * st.time_step, rng) for g in range(B)]
    x = batched_guess_block([c[2] for c in cases], [c[3] for c in cases], [c[0] for c in cases], cases[0][1], N, st.time_step,
                            model.get_total_mass())
    _, p1 = make_workload(st, model, batch=1, seed=60)
    p = np.repeat(p1, B, axis=0)
    eng = HipNlp(st, model, batch=B)
    eng.set_params(p)
    f, grad, g, jac = eng.eval(x)
    assert np.all(np.isfinite(f)) and np.all(np.isfinite(g)) and np.all(np.isfinite(jac))
    assert len({float(v) for v in f}) == B    # sixteen different guesses
    one = HipNlp(st, model, batch=1)
    one.set_params(p1)
    orc = Oracle(st, model)
    for b in range(B):
        f1, grad1, g1, jac1 = one.eval(x[b:b + 1])   # (eight-wave kernel against the batch's four-wave one: equal to the last bits)
        assert rel(f1[0], f[b]) < 1e-13 and rel(g1[0], g[b]) < 1e-13 and rel(jac1[0], jac[b]) < 1e-13 and rel(grad1[0], grad[b]) < 1e-13
        if b in (0, 7, 15):
            fo, grado, go, jaco = orc.eval(x[b], p[b])
            assert rel(f[b], fo) < TOL and rel(grad[b], grado) < TOL and rel(g[b], go) < TOL and rel(jac[b], jaco) < TOL


@pytest.mark.parametrize("name,batch,horizon", [("ramp", 2, 50), ("mixed", 1, 5), ("mixed", 120, 6)])
def test_sloped_step_tops_match_oracle(model, HipNlp, name, batch, horizon):
    """SmoothTerrain.step(top_normal_direction=...) (utilities/smooth_terrain.py:238-264; the ramp of main_walking_on_ramp.py:18-30,
    403-409): z_t = exp(-g^(2s)) pi(q_xy) with pi the plane of the given normal.  Callback quartet and exact Hessian against the oracle —
    the eight-wave kernels (small launches) and the four-wave / compact ones (batch 120), contact points on the flanks and on the top."""
    from hess_util import hess_mismatch, triplets_to_dict
    from oracle_lib import Oracle
    from test_sloped_terrain import terrains
    st = terrains(horizon, model)[name]
    x, p = make_workload(st, model, batch=batch, seed=8500 + horizon)
    place_on_step_flanks(x[:1], st, seed=horizon)
    eng, orc = HipNlp(st, model, batch=batch), Oracle(st, model)
    ir, jc = eng.sparsity()
    iro, jco = orc.sparsity()
    assert np.array_equal(ir, iro) and np.array_equal(jc, jco)
    eng.set_params(p)
    f, grad, g, jac = eng.eval(x)
    lam = np.random.RandomState(3).standard_normal((batch, orc.m))
    hr, hc = eng.hess_sparsity()
    hv = eng.eval_hess(x, 0.7, lam)
    for b in sorted({0, batch - 1, batch // 2}):
        fo, grado, go, jaco = orc.eval(x[b], p[b])
        assert rel(f[b], fo) < TOL and rel(grad[b], grado) < TOL and rel(g[b], go) < TOL and rel(jac[b], jaco) < TOL, b
        err, where = hess_mismatch(triplets_to_dict(hr, hc, hv[b]), triplets_to_dict(*orc.hess(x[b], p[b], 0.7, lam[b])), diag_scaled=True)
        assert err <= 1e-9, (b, where)
    # the slope is there: the same steps with flat tops give other rows
    flat = terrains(horizon, model)[name]
    flat.terrain_steps = [dict(t, top_normal_direction=None) for t in flat.terrain_steps]
    ef = HipNlp(flat, model, batch=batch)
    ef.set_params(p)
    assert np.max(np.abs(ef.eval(x, want=("g",))[2] - g)) > 1e-3
    # a varying-first handle through host arrays and two shards of the card: the same bits as the plain handle
    if batch == 1:
        multi = HipNlp(st, model, jac_varying_first=True, devices=[0, 0])
        plain = HipNlp(st, model, jac_varying_first=True)
        for e in (multi, plain):
            e.set_params(p)
        assert all(np.array_equal(u, v) for u, v in zip(multi.eval(x), plain.eval(x)))
        assert np.array_equal(multi.eval_hess(x, 0.7, lam), plain.eval_hess(x, 0.7, lam))


@pytest.mark.parametrize("maker,horizon,batch,vary_first", [(periodic_step_settings, 100, 1, False), (periodic_step_settings, 100, 1, True), (stairs_settings, 50, 2, False),
                                                            (single_step_settings, 30, 4, True)])
def test_two_workgroups_per_knot_give_the_one_workgroup_bits(model, maker, horizon, batch, vary_first):
    """Workgroup specialisation (SPLIT instantiations of hipnlp_knot_kernel; knot_body.h "Workgroup specialisation"): device-resident
    launches that leave half the CUs idle run the kinematic and the model-free half of the knot program on two workgroups per knot.
    Scheduling only (the reference's list is unchanged: turnkey_planners/humanoid_kinodynamic/planner.py:124-176): f, grad f, g, jac g
    are, bit for bit, those of the one-workgroup kernel (the diagnostic build's HIPNLP_SPLIT=0) — and those of the oracle to 1e-11."""
    import torch
    from diag_lib import diag_library, diag_overrides
    from hippopt_amd.hipnlp import HipNlp
    from oracle_lib import Oracle
    st = maker(horizon, model)
    x, p = make_workload(st, model, batch=batch, seed=8900 + horizon)
    if st.terrain != 0:
        place_on_step_flanks(x[:1], st, seed=3)
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x).to(dev)
    outs = {}
    for split in (1, 0):
        with diag_overrides(HIPNLP_SPLIT=split):
            eng = HipNlp(st, model, batch=batch, jac_varying_first=vary_first, library=diag_library())
        eng.set_params(p)
        o = [torch.full((batch * k,), float("nan"), dtype=torch.float64, device=dev) for k in (1, eng.n, eng.m, eng.nnz)]
        for _ in range(3):      # (the jac buffer of a varying-first handle receives its constants at first sight, the varying runs ever after)
            eng.eval_device(xd.data_ptr(), *[t.data_ptr() for t in o])
        torch.cuda.synchronize()
        outs[split] = [t.cpu().numpy().reshape(batch, -1) for t in o]
        names, terms = eng.cost_terms()
        outs[split].append(terms)
        eng.close()
    for name, a, b in zip(("f", "grad", "g", "jac", "cost terms"), outs[1], outs[0]):
        assert not np.isnan(a).any() and np.array_equal(a.view(np.uint64), b.view(np.uint64)), name
    orc = Oracle(st, model)
    ir, jc = orc.sparsity()
    fo, grado, go, jaco = orc.eval(x[0], p[0])
    f, grad, g, jac = (v[0] for v in outs[1][:4])
    assert rel(f[0], fo) < TOL and rel(grad, grado) < TOL and rel(g, go) < TOL
    if not vary_first:
        assert rel(jac, jaco) < TOL
