"""ForwardEuler / ImplicitTrapezoid against the closed forms the reference's own tests use
(test/test_integrators.py:69-117, test/test_multiple_shooting.py:253-353)."""
import numpy as np
import pytest

from hippopt_amd.integrators import ForwardEuler, ImplicitTrapezoid, multiple_shooting_defects, step


def test_one_step_of_linear_dynamics_vs_exponential():
    # reference: x_dot = lambda * x with lambda = -1, dt = 1e-3, expected x0 * exp(lambda dt) to rel 1e-4
    lam, dt, x0 = -1.0, 1e-3, 1.0

    def dyn(v, t):
        return {"x": lam * v["x"]}
    euler = step(ForwardEuler, dyn, {"x": x0}, {"x": x0}, dt)
    trap = step(ImplicitTrapezoid, dyn, {"x": x0}, {"x": x0}, dt)   # the reference passes xf == x0 too: trapezoid degenerates to Euler
    assert euler["x"] == pytest.approx(np.exp(lam * dt), rel=1e-4)
    assert trap["x"] == pytest.approx(np.exp(lam * dt), rel=1e-4)
    assert euler["x"] == trap["x"]
    # with the true end state the trapezoid rule is second order
    xf = x0 * np.exp(lam * dt)
    assert step(ImplicitTrapezoid, dyn, {"x": x0}, {"x": xf}, dt)["x"] == pytest.approx(xf, rel=1e-9)


def test_time_augmented_multi_output_dynamics():
    # reference test_integrators.py:120-151: x_dot = lambda x + t, second output discarded
    lam, dt, t0 = 0.5, 0.01, 2.0

    def dyn(v, t):
        return {"x": lam * v["x"] + t}
    out = step(ForwardEuler, dyn, {"x": 3.0}, {"x": 3.0}, dt, t0)
    assert out["x"] == pytest.approx(3.0 + dt * (lam * 3.0 + t0))
    out = step(ImplicitTrapezoid, dyn, {"x": 3.0}, {"x": 3.1}, dt, t0)
    assert out["x"] == pytest.approx(3.0 + 0.5 * dt * ((lam * 3.0 + t0) + (lam * 3.1 + t0 + dt)))


def test_toy_ocp_euler_rollout_has_zero_defects():
    """The falling-mass OCP of test_multiple_shooting.py: horizon 100, dt 0.01, x0 = 1, v0 = 0, g = -9.81; its solution is the
    explicit Euler roll-out (:336-353).  The roll-out must satisfy every ForwardEuler defect row exactly."""
    horizon, dt, g = 100, 0.01, -9.81
    traj, x, v = [], 1.0, 0.0
    for _ in range(horizon):
        traj.append({"x": x, "v": v})
        x, v = x + dt * v, v + dt * g

    def dyn(s, t):
        return {"x": s["v"], "v": g}
    defects = multiple_shooting_defects(ForwardEuler, dyn, traj, dt)
    assert len(defects) == 2 * (horizon - 1)
    assert "x[1]" in defects and "v[99]" in defects
    assert max(abs(d) for d in defects.values()) < 1e-15
    # the same trajectory violates the trapezoid defects by O(dt^2 g / 2)
    trap = multiple_shooting_defects(ImplicitTrapezoid, dyn, traj, dt)
    assert abs(trap["x[1]"]) == pytest.approx(0.5 * dt * dt * abs(g), rel=1e-9)


def test_kernel_trapezoid_rows_equal_the_host_formula(model):
    """The defect rows the engine evaluates (through the host emulation of the knot program) are the ImplicitTrapezoid formula."""
    from hippopt_amd.kinodyn_settings import single_step_settings
    from hippopt_amd.synthetic import make_workload
    from hostemu_lib import HostEmu
    st = single_step_settings(4, model)
    x, p = make_workload(st, model, 1, 5)
    e = HostEmu(st, model)
    _, _, g, _, _ = e.eval(x[0], p[0])
    name, first, rows, k0, nk = {b[0]: b for b in e.row_blocks()}["joint_position_dynamics"]
    traj = [{"s": x[0][189 * k + 157:189 * k + 180], "sd": x[0][189 * k + 134:189 * k + 157]} for k in range(4)]
    defects = multiple_shooting_defects(ImplicitTrapezoid, lambda v, t: {"s": v["sd"]}, traj, st.time_step)
    for k in range(1, 4):
        assert np.allclose(g[first + rows * (k - 1):first + rows * k], defects[f"s[{k}]"], atol=1e-15)
