"""Single-step integrators of the multiple-shooting transcription, numeric form.

Reference: integrators/forward_euler.py:24-35, integrators/implicit_trapezoid.py:24-39, dispatched by
base/single_step_integrator.py:28-37 (`step(cls, dynamics, x0, xf, dt, t0)`).  The reference builds CasADi graphs; here
`dynamics(variables: dict, time) -> dict` is any callable returning the state derivatives by name, so the same two formulas
can be checked against the reference's closed-form tests and are the host-side statement of what the kernels hard-code
(`t_points_vec`, `t_dyn`, `t_hdyn` in hippopt_amd/csrc/knot_body.h use the trapezoid rule)."""
import abc


class SingleStepIntegrator(abc.ABC):
    def __init__(self, dynamics):
        self._f = dynamics

    @classmethod
    def create(cls, dynamics):
        return cls(dynamics)

    @abc.abstractmethod
    def step(self, x0: dict, xf: dict, dt, t0=0.0) -> dict:
        ...


class ForwardEuler(SingleStepIntegrator):
    def step(self, x0, xf, dt, t0=0.0):  # xf is not used (forward_euler.py:27)
        f = self._f(x0, t0)
        return {name: x0[name] + dt * f[name] for name in f}


class ImplicitTrapezoid(SingleStepIntegrator):
    def step(self, x0, xf, dt, t0=0.0):
        f_initial = self._f(x0, t0)
        f_final = self._f(xf, t0 + dt)
        return {name: x0[name] + 0.5 * dt * (f_initial[name] + f_final[name]) for name in f_initial}


def step(cls, dynamics, x0, xf, dt, t0=0.0):
    return cls.create(dynamics).step(x0=x0, xf=xf, dt=dt, t0=t0)


def multiple_shooting_defects(cls, dynamics, trajectory: list, dt, t0=0.0):
    """Defect rows `x_{i+1} - step(x_i, x_{i+1})` for i = 0..n-2, named like the reference (`name[i+1]`,
    base/multiple_shooting_solver.py:713-742)."""
    out = {}
    for i in range(len(trajectory) - 1):
        integrated = step(cls, dynamics, trajectory[i], trajectory[i + 1], dt, t0 + i * dt)
        for name, value in integrated.items():
            out[f"{name}[{i + 1}]"] = trajectory[i + 1][name] - value
    return out
