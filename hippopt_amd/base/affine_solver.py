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

from .affine import Affine, Relation, SumOfSquares, Symbol
from .horizon import extend_structure_to_horizon
from .optimal_control import OptimizationSolver
from .optimization_object import STORAGE_TYPE, OptimizationObject


class AffineFailure(Exception):
    def __init__(self, message):
        super().__init__("The QP solver failed to solve the problem. Message: " + str(message))


class AffineSolver(OptimizationSolver):
    accepts_expressions = True

    def __init__(self, options=None):
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
        values, meta = expanded.to_dicts()
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
        self._objects.from_dict({name: self.symbol(name) for name in list(self._var) + list(self._par)})
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
        for name, value in initial_guess.to_dict().items():
            if value is None:
                continue
            target = self._var.get(name) or self._par.get(name)
            if target is None:
                continue
            arr = np.asarray(value, float)
            if arr.size != target[1]:
                raise ValueError(f"The guess for {name} has {arr.size} entries, expected {target[1]}")
            update[name] = arr.reshape(target[2])
        self._guess.from_dict(update)

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
        if not isinstance(input_cost, (SumOfSquares, Affine)):
            raise ValueError("AffineSolver costs are sums of squares of affine expressions (or linear expressions)")
        self._costs[self._fresh_name(self._costs, name, "cost_")] = input_cost

    def add_constraint(self, input_constraint, name=None):
        if not isinstance(input_constraint, Relation):
            raise ValueError("AffineSolver constraints are relations (==, <=, >=) between affine expressions")
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

    def solve(self):
        from scipy.optimize import LinearConstraint, minimize
        from scipy.sparse import csr_matrix, vstack
        flat = self._guess.to_dict()
        parameters = {}
        for name in self._par:
            if flat.get(name) is None:
                raise ValueError("The parameter " + name + " has no value.")
            parameters[name] = np.asarray(flat[name], float).reshape(-1)
        x0 = np.zeros(self._n)
        for name, (off, size, _) in self._var.items():
            if flat.get(name) is not None:
                x0[off:off + size] = np.asarray(flat[name], float).reshape(-1)
        # cost: sum_i s_i |A_i x + b_i|^2 + c^T x  ->  1/2 x^T H x + q^T x + r
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
            else:
                A, b = self._matrix(cost, parameters)
                q += np.asarray(A.sum(axis=0)).reshape(-1)
                r0 += float(b.sum())
                cost_parts[name] = ("lin", A, b, 1.0)
        H = H.tocsc()
        blocks, lo, hi, spans = [], [], [], {}
        at = 0
        for name, rel in self._constraints.items():
            A, b = self._matrix(rel.difference, parameters)
            blocks.append(A)
            lo.append(-b if rel.kind == "eq" else np.full(len(b), -np.inf))
            hi.append(-b)
            spans[name] = (at, at + len(b))
            at += len(b)
        constraints = [LinearConstraint(vstack(blocks).tocsr(), np.concatenate(lo), np.concatenate(hi))] if blocks else []
        res = minimize(lambda x: 0.5 * float(x @ (H @ x)) + float(q @ x) + r0, x0, jac=lambda x: H @ x + q, hess=lambda x: H,
                       constraints=constraints, method="trust-constr",
                       options={"maxiter": int(self._options.get("max_iter", 500)), "gtol": float(self._options.get("tol", 1e-10)),
                                "xtol": float(self._options.get("xtol", 1e-12)), "verbose": int(self._options.get("verbose", 0))})
        if res.status not in (1, 2):
            raise AffineFailure(res.message)
        x = res.x
        values = copy.deepcopy(self._guess)
        values.from_dict({name: x[off:off + size].reshape(shape) for name, (off, size, shape) in self._var.items()})
        self._values = values
        self._cost_value = float(res.fun)
        self._cost_values = {}
        for name, (kind, A, b, scale) in cost_parts.items():
            e = A @ x + b
            self._cost_values[name] = float(scale * (e @ e)) if kind == "sq" else float(e.sum())
        lam = res.v[0] if len(res.v) else np.zeros(0)
        self._multipliers = {name: np.asarray(lam[a:b_]).reshape(-1, 1) for name, (a, b_) in spans.items()}

    def get_values(self):
        return self._values

    def get_cost_value(self):
        return self._cost_value

    def get_cost_values(self):
        return self._cost_values

    def get_constraint_multipliers(self):
        return self._multipliers
