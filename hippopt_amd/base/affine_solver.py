"""`AffineSolver` — an `OptimizationSolver` plugin for problems whose constraints are affine and whose costs are sums of squares of
affine forms (plus linear terms): the closed-form OCPs of the reference's own tests (BASELINE config 1,
`test/test_multiple_shooting.py:253-353`; `test/test_optimization_problem.py`).  CPU only: a sparse convex QP handed to SciPy's
`trust-constr` with exact derivatives.  It exists so that the planner-level facade (`OptimalControlProblem`,
`MultipleShootingSolver`) can be exercised end to end — expressions, horizon transcription, named costs and multipliers — without
CasADi; the kinodynamic hot path never goes through it (`HipNlpSolver` evaluates that on the GPU).

Stands where `OptiSolver` stands in the reference (base/opti_solver.py:49-638): one decision-vector slice per variable leaf in
flatten order, parameters substituted at solve time, values returned as a filled copy of the structure, one multiplier array per
named constraint."""
import copy

import numpy as np

from .affine import Affine, Quadratic, Relation, SumOfSquares, Symbol
from .horizon import extend_structure_to_horizon
from .opti_callback import CallbackCriterion, IterateInfo, SaveBestUnsolvedVariablesCallback
from .optimal_control import OptimizationSolver
from .optimization_object import STORAGE_TYPE, OptimizationObject


def _flat(structure):
    """(values, metadata) by flattened name of an object — or of a LIST of objects, whose leaves are named `[i].<name>` as
    OptiSolver names them (base/opti_solver.py:251-333: a list input structure is generated element by element)"""
    if isinstance(structure, list):
        values, meta = {}, {}
        for i, element in enumerate(structure):
            v, m = element.to_dicts(prefix="[%d]." % i)
            values.update(v)
            meta.update(m)
        return values, meta
    return structure.to_dicts()


def _fill(structure, flat):
    if isinstance(structure, list):
        for i, element in enumerate(structure):
            element.from_dict(flat, prefix="[%d]." % i)
    else:
        structure.from_dict(flat)


class AffineFailure(Exception):
    """Counterpart of OptiFailure (base/opti_solver.py:28-37) for this plugin."""

    def __init__(self, message, callback_used=False):
        info = " and the callback did not manage to save an intermediate solution" if callback_used else ""
        super().__init__("The QP solver failed to solve the problem" + info + ". Message: " + str(message))


class AffineSolver(OptimizationSolver):
    accepts_expressions = True

    def __init__(self, options=None, callback_criterion: CallbackCriterion = None, callback_save_costs=True, callback_save_constraint_multipliers=True):
        """callback_*: as OptiSolver's (base/opti_solver.py:105-131) — with a criterion every iterate is shown to a
        SaveBestUnsolvedVariablesCallback, and a solve that fails hands back the iterate it saved instead of raising (:479-520)."""
        self._callback_criterion = callback_criterion
        self._callback_save_costs, self._callback_save_constraint_multipliers = callback_save_costs, callback_save_constraint_multipliers
        self._options = dict(options or {})
        self._structure = self._objects = self._guess = self._problem = None
        self._var, self._par = {}, {}          # leaf name -> (offset, size, shape)
        self._n = 0
        self._costs, self._constraints = {}, {}
        self._values = self._cost_value = None
        self._cost_values, self._multipliers = {}, {}

    # ---- structure ----------------------------------------------------------------------------------------------------------
    def generate_optimization_objects(self, input_structure, **kwargs):
        if not isinstance(input_structure, (OptimizationObject, list)):
            raise ValueError("The input structure is neither an optimization object nor a list.")
        self._structure = copy.deepcopy(input_structure)
        expanded = extend_structure_to_horizon(input_structure, **kwargs)
        values, meta = _flat(expanded)
        at = 0
        for name, value in values.items():
            if not isinstance(value, np.ndarray) or value.ndim != 2 or value.size == 0:
                raise ValueError("Field " + name + " is tagged as storage, but it is not a non-empty 2-D array.")
            if meta[name][STORAGE_TYPE] == "variable":
                self._var[name] = (at, value.size, value.shape)
                at += value.size
            else:
                self._par[name] = (0, value.size, value.shape)
        self._n = at
        self._guess = copy.deepcopy(expanded)          # numeric values: initial guess of the variables, values of the parameters
        self._symbols = {}
        self._objects = copy.deepcopy(expanded)        # the same tree with an expression at every leaf (what OptiSolver's MX tree is)
        _fill(self._objects, {name: self.symbol(name) for name in list(self._var) + list(self._par)})
        return self._objects

    def get_optimization_objects(self):
        return self._objects

    def get_optimization_structure(self):
        return self._structure

    def register_problem(self, problem):
        self._problem = problem

    def get_problem(self):
        return self._problem

    def symbol(self, leaf_name):
        """expression of one leaf of the EXPANDED structure (a knot of a time-varying variable, or a constant)"""
        info = self._var.get(leaf_name) or self._par.get(leaf_name)
        if info is None:
            raise ValueError("Variable " + leaf_name + " not found in the optimization variables.")
        if leaf_name not in self._symbols:             # (one object per leaf: `initial(g) is final(g)` for a constant)
            self._symbols[leaf_name] = Symbol(leaf_name, info[1])
        return self._symbols[leaf_name]

    def symbolic_structure(self, input_structure, names):
        """a copy of the un-expanded structure whose storage leaves are symbols named by their flattened (time-generic) names"""
        sym = copy.deepcopy(input_structure)
        flat = sym.to_dict()
        sym.from_dict({name: Symbol(name, int(np.size(value))) for name, value in flat.items()})
        return sym

    # ---- guesses ---------------------------------------------------------------------------------------------------------------
    def set_initial_guess(self, initial_guess):
        update = {}
        for name, value in _flat(initial_guess)[0].items():
            if value is None:
                continue
            target = self._var.get(name) or self._par.get(name)
            if target is None:
                continue
            arr = np.asarray(value, float)
            if arr.size != target[1]:
                raise ValueError(f"The guess for {name} has {arr.size} entries, expected {target[1]}")
            update[name] = arr.reshape(target[2])
        _fill(self._guess, update)

    def get_initial_guess(self):
        return copy.deepcopy(self._guess)

    # ---- description -------------------------------------------------------------------------------------------------------------
    def _fresh_name(self, table, name, prefix):
        if name is None:
            name = prefix + str(len(table))
        if name in table:
            raise ValueError("The name " + name + " is already used.")
        return name

    def add_cost(self, input_cost, name=None):
        if not isinstance(input_cost, (SumOfSquares, Affine, Quadratic)):
            raise ValueError("AffineSolver costs are sums of squares of affine expressions, quadratic or linear expressions")
        self._costs[self._fresh_name(self._costs, name, "cost_")] = input_cost

    def add_constraint(self, input_constraint, name=None):
        if not isinstance(input_constraint, Relation):
            raise ValueError("AffineSolver constraints are relations (==, <=, >=) between affine (or quadratic) expressions")
        self._constraints[self._fresh_name(self._constraints, name, "constraint_")] = input_constraint

    def cost_function(self):
        return dict(self._costs)

    def get_cost_expressions(self):
        return dict(self._costs)

    def get_constraint_expressions(self):
        return dict(self._constraints)

    # ---- solve ---------------------------------------------------------------------------------------------------------------------
    def _matrix(self, expression, parameters):
        """(A csr [rows x n], b [rows]) of an affine expression with the parameters substituted"""
        from scipy.sparse import csr_matrix
        rows, cols, vals = [], [], []
        b = expression.const.copy()
        for r, row in enumerate(expression.rows):
            for (leaf, i), c in row.items():
                if leaf in self._var:
                    rows.append(r); cols.append(self._var[leaf][0] + i); vals.append(c)   # noqa: E702
                elif leaf in parameters:
                    b[r] += c * parameters[leaf][i]
                else:
                    raise ValueError("unknown leaf " + leaf + " in an expression")
        return csr_matrix((vals, (rows, cols)), shape=(len(expression), self._n)), b

    def _quadratic(self, expression, parameters):
        """row r of a quadratic expression as x^T Q_r x / 2 + a_r^T x + b_r with the parameters substituted:
        ([Q_r csr, symmetric], A csr [rows x n], b [rows])"""
        from scipy.sparse import csr_matrix
        A, b = self._matrix(expression.affine, parameters)
        A = A.tolil()
        Qs = []
        for r, q in enumerate(expression.quad):
            rows, cols, vals = [], [], []
            for (ka, kb), c in q.items():
                va, vb = ka[0] in self._var, kb[0] in self._var
                for k, known in ((ka, va), (kb, vb)):
                    if not known and k[0] not in parameters:
                        raise ValueError("unknown leaf " + k[0] + " in an expression")
                if va and vb:
                    i, j = self._var[ka[0]][0] + ka[1], self._var[kb[0]][0] + kb[1]
                    rows += [i, j]; cols += [j, i]; vals += [c, c]                        # noqa: E702  (x_i x_j = x^T (e_i e_j^T + e_j e_i^T) x / 2)
                elif va or vb:                                                           # variable x parameter: linear
                    kv, kp = (ka, kb) if va else (kb, ka)
                    A[r, self._var[kv[0]][0] + kv[1]] += c * parameters[kp[0]][kp[1]]
                else:
                    b[r] += c * parameters[ka[0]][ka[1]] * parameters[kb[0]][kb[1]]
            Qs.append(csr_matrix((vals, (rows, cols)), shape=(self._n, self._n)))
        return Qs, A.tocsr(), b

    def solve(self):
        from scipy.optimize import LinearConstraint, NonlinearConstraint, minimize
        from scipy.sparse import csr_matrix, vstack
        flat = _flat(self._guess)[0]
        parameters = {}
        for name in self._par:
            if flat.get(name) is None:
                raise ValueError("The parameter " + name + " has no value.")
            parameters[name] = np.asarray(flat[name], float).reshape(-1)
        x0 = np.zeros(self._n)
        for name, (off, size, _) in self._var.items():
            if flat.get(name) is not None:
                x0[off:off + size] = np.asarray(flat[name], float).reshape(-1)
        # cost: sum_i s_i |A_i x + b_i|^2 + quadratic and linear terms  ->  1/2 x^T H x + q^T x + r
        H = csr_matrix((self._n, self._n))
        q, r0 = np.zeros(self._n), 0.0
        cost_parts = {}
        for name, cost in self._costs.items():
            if isinstance(cost, SumOfSquares):
                A, b = self._matrix(cost.expression, parameters)
                H = H + 2.0 * cost.scaling * (A.T @ A)
                q += 2.0 * cost.scaling * (A.T @ b)
                r0 += cost.scaling * float(b @ b)
                cost_parts[name] = ("sq", A, b, cost.scaling)
            elif isinstance(cost, Quadratic):
                Qs, A, b = self._quadratic(cost, parameters)
                Qsum = sum(Qs[1:], Qs[0]) if Qs else csr_matrix((self._n, self._n))
                H = H + Qsum
                q += np.asarray(A.sum(axis=0)).reshape(-1)
                r0 += float(b.sum())
                cost_parts[name] = ("quad", (Qsum, A), b, 1.0)
            else:
                A, b = self._matrix(cost, parameters)
                q += np.asarray(A.sum(axis=0)).reshape(-1)
                r0 += float(b.sum())
                cost_parts[name] = ("lin", A, b, 1.0)
        H = H.tocsc()
        # constraints: the affine rows as ONE LinearConstraint; every quadratic relation a NonlinearConstraint of its own (the named
        # multipliers come back per constraint either way)
        blocks, lo, hi, spans = [], [], [], {}
        nonlinear = []     # (name, NonlinearConstraint, rows)
        at = 0
        for name, rel in self._constraints.items():
            if isinstance(rel.difference, Quadratic):
                Qs, A, b = self._quadratic(rel.difference, parameters)

                def fun(x, Qs=Qs, A=A, b=b):
                    return np.array([0.5 * float(x @ (Q @ x)) for Q in Qs]) + A @ x + b

                def jac(x, Qs=Qs, A=A):
                    return vstack([csr_matrix(Q @ x) for Q in Qs]) + A

                def hess(x, v, Qs=Qs):
                    return sum((vi * Q for vi, Q in zip(v, Qs)), csr_matrix((self._n, self._n)))
                rows = len(b)
                upper = np.zeros(rows)
                lower = np.zeros(rows) if rel.kind == "eq" else np.full(rows, -np.inf)
                nonlinear.append((name, NonlinearConstraint(fun, lower, upper, jac=jac, hess=hess), rows))
                continue
            A, b = self._matrix(rel.difference, parameters)
            blocks.append(A)
            lo.append(-b if rel.kind == "eq" else np.full(len(b), -np.inf))
            hi.append(-b)
            spans[name] = (at, at + len(b))
            at += len(b)
        constraints = [LinearConstraint(vstack(blocks).tocsr(), np.concatenate(lo), np.concatenate(hi))] if blocks else []
        constraints += [c for _, c, _ in nonlinear]

        def cost_of(x):
            return 0.5 * float(x @ (H @ x)) + float(q @ x) + r0

        def cost_values_at(x):
            out = {}
            for name, (kind, A, b, scale) in cost_parts.items():
                if kind == "quad":
                    Qsum, Aq = A
                    out[name] = float(0.5 * (x @ (Qsum @ x)) + (Aq @ x + b).sum())
                else:
                    e = A @ x + b
                    out[name] = float(scale * (e @ e)) if kind == "sq" else float(e.sum())
            return out

        def multipliers_of(v):
            lam = {}
            vs = list(v)
            if blocks:
                lin = np.asarray(vs.pop(0))
                lam.update({name: lin[a:b_].reshape(-1, 1) for name, (a, b_) in spans.items()})
            for (name, _, rows), vi in zip(nonlinear, vs):
                lam[name] = np.asarray(vi).reshape(-1, 1)
            return lam
        saver = None
        if self._callback_criterion is not None:      # opti_solver.py:451-477
            saver = SaveBestUnsolvedVariablesCallback(self._callback_criterion, self._callback_save_costs, self._callback_save_constraint_multipliers)
        saved = {}

        def on_iterate(xk, state):
            if saver is None:
                return False
            before = saver.best_iteration
            saver(IterateInfo(int(state.nit), float(state.fun), float(state.constr_violation)), xk, None,
                  (lambda: cost_values_at(np.asarray(xk, float))) if self._callback_save_costs else None)
            if saver.best_iteration != before and self._callback_save_constraint_multipliers:
                saved["lam"] = multipliers_of(state.v)
            return False
        res = minimize(cost_of, x0, jac=lambda x: H @ x + q, hess=lambda x: H, constraints=constraints, method="trust-constr", callback=on_iterate,
                       options={"maxiter": int(self._options.get("max_iter", 500)), "gtol": float(self._options.get("tol", 1e-10)),
                                "xtol": float(self._options.get("xtol", 1e-12)), "verbose": int(self._options.get("verbose", 0)),
                                # (inequalities go through trust-constr's barrier method: its default final barrier parameter, 1e-8,
                                #  leaves an active bound 1e-6 off)
                                "barrier_tol": float(self._options.get("barrier_tol", 1e-12))})
        # what IPOPT reports as infeasible / not converged makes Opti raise, and OptiSolver.solve turn that into OptiFailure unless the
        # callback saved an iterate (opti_solver.py:479-520)
        violation = float(res.constr_violation) if constraints else 0.0
        failure = None
        if res.status not in (1, 2):
            failure = res.message
        elif violation > float(self._options.get("constr_viol_tol", 1e-6)):
            failure = "converged to a point that violates the constraints by %.3g (infeasible problem)" % violation
        x, lam = res.x, (multipliers_of(res.v) if len(res.v) else {})
        if failure is None and not nonlinear:
            # A convex QP with linear rows: trust-constr's barrier iterate is within ~1e-6 of the optimum (it stops on the gradient norm,
            # whatever its barrier parameter); the active set it has found gives the optimum itself by one KKT solve
            polished = self._polish(H, q, vstack(blocks).tocsr() if blocks else None, np.concatenate(lo) if blocks else None,
                                    np.concatenate(hi) if blocks else None, x)
            if polished is not None:
                x, v_lin = polished
                lam = multipliers_of([v_lin]) if blocks else {}
        if failure is not None:
            if saver is None or saver.best_iteration is None:
                raise AffineFailure(failure, callback_used=saver is not None)
            x, lam = saver.best_x, saved.get("lam", {})
        values = copy.deepcopy(self._guess)
        _fill(values, {name: x[off:off + size].reshape(shape) for name, (off, size, shape) in self._var.items()})
        self._values = values
        self._cost_value = cost_of(x)
        self._cost_values = cost_values_at(x)
        self._multipliers = lam
        self._solve_info = {"status": int(res.status), "iterations": int(res.nit), "constr_violation": violation, "failure": failure,
                            "used_saved_iterate": failure is not None}

    @staticmethod
    def _polish(H, q, A, lo, hi, x):
        """the exact optimum of  min x^T H x / 2 + q^T x  s.t.  A x = hi on the equality rows and on the inequality rows (A x <= hi) that are
        active at x: one sparse KKT solve, accepted only if the result is feasible and the multipliers of the active inequalities have the
        right sign (otherwise None: the interior-point iterate stands).  Returns (x, multipliers of all rows)."""
        from scipy.sparse import bmat, csr_matrix
        from scipy.sparse.linalg import spsolve
        n = H.shape[0]
        try:
            if A is None:
                xs = spsolve(H.tocsc(), -q)
                return (xs, np.zeros(0)) if np.all(np.isfinite(xs)) else None
            r = A @ x
            eq = lo == hi
            act = eq | (hi - r <= 1e-5 * (1.0 + np.abs(hi)))
            idx = np.nonzero(act)[0]
            Aa = A[idx]
            K = bmat([[H, Aa.T], [Aa, csr_matrix((idx.size, idx.size))]], format="csc")
            sol = spsolve(K, np.concatenate([-q, hi[idx]]))
            if not np.all(np.isfinite(sol)):
                return None
            xs, la = sol[:n], sol[n:]
            rs = A @ xs
            if np.any(rs - hi > 1e-9 * (1.0 + np.abs(hi))) or np.any((~eq[idx]) & (la < -1e-9)):
                return None
            if np.linalg.norm(xs - x) > 1e-3 * (1.0 + np.linalg.norm(x)):     # (not the point the solver found: leave it alone)
                return None
            v = np.zeros(A.shape[0])
            v[idx] = la
            return xs, v
        except Exception:  # noqa: BLE001  (a singular KKT matrix — redundant active rows: no polish)
            return None

    def get_values(self):
        return self._values

    def get_cost_value(self):
        return self._cost_value

    def get_cost_values(self):
        return self._cost_values

    def get_constraint_multipliers(self):
        return self._multipliers
