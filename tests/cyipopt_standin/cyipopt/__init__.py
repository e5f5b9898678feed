"""TEST-ONLY stand-in for the part of cyipopt's API that hippopt_amd/hipnlp_solver.py binds (cyipopt and IPOPT are not in the image):
`Problem(n, m, problem_obj, lb, ub, cl, cu)`, `add_option`, `solve(x0) -> (x, info)`, `get_current_iterate()`, and the callback
protocol of `problem_obj` — objective / gradient / constraints / jacobianstructure / jacobian / hessianstructure / hessian /
intermediate — with cyipopt's argument lists and info keys.  The iterations themselves are SciPy trust-constr: this exercises the
adapter's call plumbing (what is asked of the engine, in which order, with which shapes), not IPOPT's algorithm."""
import numpy as np
from scipy.optimize import BFGS, Bounds, NonlinearConstraint, minimize
from scipy.sparse import csc_matrix

__version__ = "0.0-standin"


class CyIpoptEvaluationError(ArithmeticError):
    pass


class Problem:
    def __init__(self, n, m, problem_obj=None, lb=None, ub=None, cl=None, cu=None):
        self.n, self.m, self.obj = int(n), int(m), problem_obj
        self.lb = np.full(n, -np.inf) if lb is None else np.asarray(lb, float)
        self.ub = np.full(n, np.inf) if ub is None else np.asarray(ub, float)
        self.cl, self.cu = np.asarray(cl, float), np.asarray(cu, float)
        self.options = {}
        self._iterate = None
        for name in ("objective", "gradient", "constraints", "jacobian"):
            if not callable(getattr(problem_obj, name, None)):
                raise TypeError("problem_obj lacks " + name)

    def add_option(self, key, value):
        if not isinstance(key, str) or isinstance(value, (list, dict)):
            raise TypeError("invalid option")
        self.options[key] = value

    def get_current_iterate(self, scaled=False):
        if self._iterate is None:
            raise RuntimeError("only callable from intermediate()")
        return self._iterate

    def solve(self, x0, lagrange=None, zl=None, zu=None):
        o = self.obj
        ir, jc = (np.asarray(a) for a in o.jacobianstructure())
        exact = callable(getattr(o, "hessian", None)) and self.options.get("hessian_approximation", "exact") != "limited-memory"
        if exact:
            hr, hc = (np.asarray(a) for a in o.hessianstructure())
            off = hr != hc

            def sym(vals):
                return csc_matrix((np.concatenate([vals, vals[off]]), (np.concatenate([hr, hc[off]]), np.concatenate([hc, hr[off]]))), shape=(self.n, self.n))
            zero = np.zeros(self.m)
            hess_f = lambda x: sym(np.asarray(o.hessian(x, zero, 1.0)))          # noqa: E731
            hess_c = lambda x, v: sym(np.asarray(o.hessian(x, np.asarray(v), 0.0)))   # noqa: E731
        else:
            hess_f, hess_c = BFGS(), BFGS()
        nlc = NonlinearConstraint(lambda x: np.asarray(o.constraints(x)), self.cl, self.cu,
                                  jac=lambda x: csc_matrix((np.asarray(o.jacobian(x)), (ir, jc)), shape=(self.m, self.n)), hess=hess_c)
        bounded = bool(np.any(np.isfinite(self.lb)) or np.any(np.isfinite(self.ub)))

        def cb(xk, state):
            if callable(getattr(o, "intermediate", None)):
                lam_x = state.v[1] if bounded and len(state.v) > 1 else np.zeros(self.n)
                self._iterate = {"x": np.array(xk), "mult_g": np.array(state.v[0]), "mult_x_L": np.maximum(-lam_x, 0.0), "mult_x_U": np.maximum(lam_x, 0.0)}
                go_on = o.intermediate(0, int(state.nit), float(state.fun), float(state.constr_violation), float(state.optimality), 0.0,
                                       0.0, 0.0, 1.0, 1.0, 0)
                self._iterate = None
                return go_on is False
            return False
        res = minimize(lambda x: float(o.objective(x)), np.asarray(x0, float), jac=lambda x: np.asarray(o.gradient(x)), hess=hess_f,
                       constraints=[nlc], bounds=Bounds(self.lb, self.ub) if bounded else None, method="trust-constr", callback=cb,
                       options={"maxiter": int(self.options.get("max_iter", 50)), "verbose": 0})
        lam_x = res.v[1] if bounded and len(res.v) > 1 else np.zeros(self.n)
        status = 0 if res.status in (1, 2) else (-1 if res.status == 0 else 5)   # Solve_Succeeded / Maximum_Iterations_Exceeded / User_Requested_Stop
        info = {"x": res.x, "g": np.asarray(o.constraints(res.x)), "obj_val": float(res.fun), "mult_g": np.array(res.v[0]),
                "mult_x_L": np.maximum(-lam_x, 0.0), "mult_x_U": np.maximum(lam_x, 0.0), "status": status,
                "status_msg": {0: b"Algorithm terminated successfully", -1: b"Maximum number of iterations exceeded", 5: b"User requested stop"}[status]}
        return res.x, info
