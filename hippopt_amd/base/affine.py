"""Affine expressions over the leaves of an optimisation structure, and the transcription rules of the multiple-shooting layer on them.

The reference describes every problem to its solver as CasADi graphs.  The engine-backed solver does not take expressions at all
(its problem is typed, `hippopt_amd/csrc/layout.h`); what remains are the small OCPs of the reference's own tests and examples —
affine dynamics, box constraints, quadratic costs (`test/test_multiple_shooting.py:253-353`: BASELINE config 1).  For those an
expression is a vector of affine forms  A z + b  over named leaf entries, which is everything `AffineSolver` (a sparse convex QP on
SciPy, CPU) needs; nothing here evaluates anything on a GPU.

What is restated from the reference is the MEANING of the four transcription calls, not their code:
  * `dot(x) == rhs`                       base/dynamics.py:149-251 (rhs: a function of named inputs with a name map, a list of variable
                                          names for "x_dot = y", or an expression)
  * `add_dynamics`                        base/multiple_shooting_solver.py:578-742: for i = 0 .. n - 2 the rows
                                          x[i + 1] == step(x[i], x[i + 1], dt, t0 + i dt), named `name[i + 1]{j}`; optional x0 rows;
                                          mode minimize turns every row into scaling * sumsqr(lhs - rhs) (base/problem.py:118-122)
  * `add_expression_to_horizon`           :774-824: the expression with every time-varying leaf replaced by its knot-i value, i = 0 (or 1)
                                          .. n - 1, n the shortest horizon > 1 among the leaves involved
  * `add_cost` / `add_constraint`         base/problem.py:95-174: `==` as a cost is sumsqr of the difference; a bare scalar constraint is
                                          `expr == expected_value`; generators are unrolled with `{i}` appended to the name
"""
import inspect
import types

import numpy as np

from .problem import ExpressionType


class Affine:
    """rows x 1 affine forms: row r = const[r] + sum_k coeff[r][k] * leaf entry k, a leaf entry being (leaf name, index)"""

    __array_priority__ = 1000   # numpy scalars / arrays on the left defer to the operators below

    def __init__(self, rows, const=None):
        self.rows = rows                                           # list of {(leaf, index): coefficient}
        self.const = np.zeros(len(rows)) if const is None else np.asarray(const, float).reshape(-1)

    # ---- construction ---------------------------------------------------------------------------------------------------
    @staticmethod
    def leaf(name, size):
        return Affine([{(name, i): 1.0} for i in range(size)])

    @staticmethod
    def constant(value):
        value = np.asarray(value, float).reshape(-1)
        return Affine([{} for _ in range(value.size)], value)

    @staticmethod
    def lift(value, like=None):
        if isinstance(value, Affine):
            return value
        out = Affine.constant(value)
        if like is not None and len(out) == 1 and len(like) > 1:
            out = Affine([{} for _ in range(len(like))], np.full(len(like), out.const[0]))
        return out

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, item):
        if isinstance(item, slice):
            return Affine(self.rows[item], self.const[item])
        return Affine([self.rows[item]], [self.const[item]])

    # ---- algebra ----------------------------------------------------------------------------------------------------------
    def _combine(self, other, sign):
        if isinstance(other, Quadratic):                 # affine +- quadratic
            return (other * sign)._combine(self, 1.0)
        other = Affine.lift(other, self)
        mine = self if len(self) == len(other) else Affine.lift(self, other) if len(self) == 1 else self
        if len(mine) == 1 and len(other) > 1:
            mine = Affine([dict(mine.rows[0]) for _ in range(len(other))], np.full(len(other), mine.const[0]))
        if len(mine) != len(other):
            raise ValueError(f"affine expressions of {len(mine)} and {len(other)} rows do not combine")
        rows = []
        for a, b in zip(mine.rows, other.rows):
            r = dict(a)
            for k, v in b.items():
                r[k] = r.get(k, 0.0) + sign * v
            rows.append(r)
        return Affine(rows, mine.const + sign * other.const)

    def __add__(self, other):
        return self._combine(other, 1.0)

    __radd__ = __add__

    def __sub__(self, other):
        return self._combine(other, -1.0)

    def __rsub__(self, other):
        return (-self)._combine(other, 1.0)

    def __neg__(self):
        return self * -1.0

    def __mul__(self, factor):
        if isinstance(factor, Quadratic):
            return factor * self
        if isinstance(factor, Affine):
            if any(factor.rows) and any(self.rows):
                return Quadratic.product(self, factor)      # (row by row: what `*` is between two CasADi column vectors)
            if any(factor.rows):
                return factor * self
            factor = factor.const if len(factor) > 1 else factor.const[0]
        factor = np.asarray(factor, float).reshape(-1)
        if factor.size not in (1, len(self)):
            raise ValueError("scaling of the wrong size")
        f = np.broadcast_to(factor, (len(self),))
        return Affine([{k: v * f[r] for k, v in row.items()} for r, row in enumerate(self.rows)], self.const * f)

    __rmul__ = __mul__

    def __truediv__(self, divisor):
        return self * (1.0 / np.asarray(divisor, float))

    def __pow__(self, exponent):
        if exponent == 1:
            return self
        if exponent == 2:
            return Quadratic.product(self, self)
        raise ValueError("only squares of affine expressions are supported")

    # ---- relations ---------------------------------------------------------------------------------------------------------
    def __eq__(self, other):   # noqa: PLW1641  (expressions are not hashable on purpose)
        return Relation(self - other, "eq")

    def __le__(self, other):
        return Relation(self - other, "le")

    def __ge__(self, other):
        if isinstance(other, Quadratic):
            return Relation(other - self, "le")
        return Relation(Affine.lift(other, self) - self, "le")

    __hash__ = None

    # ---- use ------------------------------------------------------------------------------------------------------------------
    def leaves(self):
        return {k[0] for row in self.rows for k in row}

    def renamed(self, mapping):
        """the same forms over other leaves: {leaf name: leaf name} (knot substitution)"""
        return Affine([{(mapping.get(k[0], k[0]), k[1]): v for k, v in row.items()} for row in self.rows], self.const.copy())

    def value(self, values):
        """numeric value given {leaf name: array}"""
        out = self.const.copy()
        for r, row in enumerate(self.rows):
            for (leaf, i), c in row.items():
                out[r] += c * np.asarray(values[leaf], float).reshape(-1)[i]
        return out


class Quadratic:
    """rows x 1 quadratic forms: row r = affine[r] + sum coeff[r][(k, l)] * entry k * entry l  (k <= l in a fixed order of the leaf entries).
    What the products of the reference's toy problems are — `a[k] * cs.power(x[k], 2) + b[k] * x[k]`, `x * x`, `(x - 5) ** 2`
    (test/test_optimization_problem.py:30-68, 194-254) — and nothing more general: a quadratic times an expression is refused."""

    __array_priority__ = 1000

    def __init__(self, quad, affine):
        self.quad = quad            # list of {((leaf, i), (leaf, j)): coefficient}
        self.affine = affine        # Affine of the same length

    @staticmethod
    def product(a, b):
        a, b = Affine.lift(a), Affine.lift(b)
        if len(a) == 1 and len(b) > 1:
            a = Affine([dict(a.rows[0]) for _ in range(len(b))], np.full(len(b), a.const[0]))
        if len(b) == 1 and len(a) > 1:
            b = Affine([dict(b.rows[0]) for _ in range(len(a))], np.full(len(a), b.const[0]))
        if len(a) != len(b):
            raise ValueError(f"expressions of {len(a)} and {len(b)} rows do not multiply row by row")
        quad, lin_rows, const = [], [], np.zeros(len(a))
        for r in range(len(a)):
            q, lin = {}, {}
            for ka, va in a.rows[r].items():
                for kb, vb in b.rows[r].items():
                    key = (ka, kb) if ka <= kb else (kb, ka)
                    q[key] = q.get(key, 0.0) + va * vb
            for k, v in a.rows[r].items():
                lin[k] = lin.get(k, 0.0) + v * b.const[r]
            for k, v in b.rows[r].items():
                lin[k] = lin.get(k, 0.0) + v * a.const[r]
            const[r] = a.const[r] * b.const[r]
            quad.append(q)
            lin_rows.append(lin)
        return Quadratic(quad, Affine(lin_rows, const))

    def __len__(self):
        return len(self.quad)

    def __getitem__(self, item):
        if isinstance(item, slice):
            return Quadratic(self.quad[item], self.affine[item])
        return Quadratic([self.quad[item]], self.affine[item])

    def _combine(self, other, sign):
        if isinstance(other, Quadratic):
            if len(other) != len(self):
                raise ValueError("quadratic expressions of different lengths do not combine")
            quad = []
            for a, b in zip(self.quad, other.quad):
                q = dict(a)
                for k, v in b.items():
                    q[k] = q.get(k, 0.0) + sign * v
                quad.append(q)
            return Quadratic(quad, self.affine._combine(other.affine, sign))
        return Quadratic([dict(q) for q in self.quad], self.affine._combine(other, sign))

    def __add__(self, other):
        return self._combine(other, 1.0)

    __radd__ = __add__

    def __sub__(self, other):
        return self._combine(other, -1.0)

    def __rsub__(self, other):
        return (-self)._combine(other, 1.0)

    def __neg__(self):
        return self * -1.0

    def __mul__(self, factor):
        if isinstance(factor, Quadratic) or (isinstance(factor, Affine) and any(factor.rows)):
            raise ValueError("the product of a quadratic expression and an expression is not quadratic")
        if isinstance(factor, Affine):
            factor = factor.const if len(factor) > 1 else factor.const[0]
        f = np.broadcast_to(np.asarray(factor, float).reshape(-1), (len(self),))
        return Quadratic([{k: v * f[r] for k, v in q.items()} for r, q in enumerate(self.quad)], self.affine * f)

    __rmul__ = __mul__

    def __truediv__(self, divisor):
        return self * (1.0 / np.asarray(divisor, float))

    def __eq__(self, other):   # noqa: PLW1641
        return Relation(self - other, "eq")

    def __le__(self, other):
        return Relation(self - other, "le")

    def __ge__(self, other):
        return Relation(-(self - other), "le")

    __hash__ = None

    def leaves(self):
        return {k[0] for q in self.quad for pair in q for k in pair} | self.affine.leaves()

    def renamed(self, mapping):
        ren = lambda k: (mapping.get(k[0], k[0]), k[1])  # noqa: E731
        return Quadratic([{tuple(sorted((ren(ka), ren(kb)))): v for (ka, kb), v in q.items()} for q in self.quad], self.affine.renamed(mapping))

    def value(self, values):
        out = self.affine.value(values)
        for r, q in enumerate(self.quad):
            for (ka, kb), c in q.items():
                out[r] += c * np.asarray(values[ka[0]], float).reshape(-1)[ka[1]] * np.asarray(values[kb[0]], float).reshape(-1)[kb[1]]
        return out


def power(expression, exponent):
    """`cs.power(x, 2)` of the reference's toy costs"""
    return Affine.lift(expression) ** exponent


class Symbol(Affine):
    """a leaf of the symbolic structure: an expression that also knows the flattened name it stands for"""

    def __init__(self, flat_name, size):
        super().__init__([{(flat_name, i): 1.0} for i in range(size)])
        self.flat_name = flat_name

    def name(self):
        return self.flat_name

    def __hash__(self):   # (symbols serve as dictionary keys — x0={symbol: value} —; general expressions do not)
        return hash(("Symbol", self.flat_name))


class Relation:
    """`difference` (= lhs - rhs) == 0 or <= 0, row by row"""

    def __init__(self, difference, kind):
        self.difference, self.kind = difference, kind

    def renamed(self, mapping):
        return Relation(self.difference.renamed(mapping), self.kind)

    def leaves(self):
        return self.difference.leaves()


class SumOfSquares:
    """scaling * sum_r expression[r]^2"""

    def __init__(self, expression, scaling=1.0):
        self.expression, self.scaling = Affine.lift(expression), float(scaling)

    def renamed(self, mapping):
        return SumOfSquares(self.expression.renamed(mapping), self.scaling)

    def leaves(self):
        return self.expression.leaves()

    def __mul__(self, factor):
        return SumOfSquares(self.expression, self.scaling * float(factor))

    __rmul__ = __mul__


def sumsqr(expression):
    return SumOfSquares(expression)


# ---- problem-level conversions (base/problem.py:95-174) -------------------------------------------------------------------------------
def _unroll(expression, name):
    if isinstance(expression, types.GeneratorType):
        for i, e in enumerate(expression):
            yield (None if name is None else name + "{" + str(i) + "}"), e
    else:
        yield name, expression


def as_cost(expression, scaling=1.0, name=None):
    for label, e in _unroll(expression, name):
        if isinstance(e, Relation):
            if e.kind != "eq":
                raise ValueError("The conversion from an inequality to a cost is not yet supported")
            if isinstance(e.difference, Quadratic):
                raise ValueError("sumsqr of a quadratic expression is a quartic cost: not supported by the affine / quadratic expressions")
            yield label, SumOfSquares(e.difference, scaling)
        elif isinstance(e, SumOfSquares):
            yield label, e * scaling
        elif isinstance(e, Quadratic):
            yield label, e * scaling                  # a quadratic cost (its rows are summed)
        else:
            yield label, Affine.lift(e) * scaling     # a linear cost


def as_constraint(expression, expected_value=0.0, name=None):
    for label, e in _unroll(expression, name):
        if isinstance(e, Relation):
            yield label, e
        else:
            e = e if isinstance(e, Quadratic) else Affine.lift(e)
            if len(e) != 1:
                raise ValueError("The input expression is not supported.")
            yield label, e == expected_value


# ---- dynamics (base/dynamics.py) --------------------------------------------------------------------------------------------------------
class DotLHS:
    def __init__(self, states, time_name="t"):
        self.states = [getattr(s, "flat_name", s) for s in (states if isinstance(states, list) else [states])]
        self.time_name = time_name

    def __eq__(self, rhs):   # noqa: PLW1641
        return Dynamics(self, rhs)

    __hash__ = None


def dot(x, t="t"):
    """`dot(x) == rhs` declares x_dot = rhs; x: a flattened name, a symbol of the symbolic structure, or a list of either"""
    return DotLHS(x, getattr(t, "flat_name", t))


class Dynamics:
    """rhs: (function, {variable name: input name}) | function | list of variable names (x_dot = y) | a name | an expression"""

    def __init__(self, lhs, rhs):
        self.lhs = lhs
        names_map = {}
        if isinstance(rhs, tuple):
            rhs, names_map = rhs
        if isinstance(rhs, (str, Symbol)):
            rhs = [getattr(rhs, "flat_name", rhs)]
        self.rhs, self.names_map = rhs, dict(names_map)

    def state_variables(self):
        return list(self.lhs.states)

    def input_names(self):
        if isinstance(self.rhs, list):
            return [getattr(v, "flat_name", v) for v in self.rhs]
        if callable(self.rhs):
            inverse = {v: k for k, v in self.names_map.items()}
            return [inverse.get(a, a) for a in inspect.signature(self.rhs).parameters]
        return sorted(Affine.lift(self.rhs).leaves())

    def evaluate(self, variables, time):
        """{state name: derivative expression} given {variable name: expression}"""
        if isinstance(self.rhs, list):
            outs = [variables[n] for n in self.input_names()]
        elif callable(self.rhs):
            inputs = {}
            for name in self.input_names():
                inputs[self.names_map.get(name, name)] = time if name == self.lhs.time_name else variables[name]
            outs = self.rhs(**inputs)
            outs = list(outs.values()) if isinstance(outs, dict) else (list(outs) if isinstance(outs, (tuple, list)) else [outs])
        else:
            outs = [Affine.lift(self.rhs).renamed({n: variables[n].flat_name for n in Affine.lift(self.rhs).leaves()
                                                   if hasattr(variables.get(n), "flat_name")})]
        if len(outs) < len(self.lhs.states):
            raise ValueError("the dynamics returns fewer outputs than there are states")
        return dict(zip(self.lhs.states, outs))   # (extra outputs are discarded, as in test_integrators.py:120-151)


# ---- the horizon (base/multiple_shooting_solver.py:578-824) ------------------------------------------------------------------------------
def _knot_symbol(ms, name, k):
    """expression of flattened variable `name` at knot k (constants — horizon 1 — are the same at every knot)"""
    names = ms._knot_names(name)
    return ms.get_optimization_solver().symbol(names[min(k, len(names) - 1)])


def add_dynamics_to_horizon(ms, dynamics, x0=None, t0=0.0, mode=None, name=None, x0_name=None, default_integrator=None, **kwargs):
    from .. import integrators
    mode = ExpressionType.subject_to if mode is None else mode
    if not isinstance(dynamics, Dynamics):
        raise ValueError("add_dynamics wants `dot(x) == rhs`")
    if "dt" not in kwargs:
        raise ValueError("MultipleShootingSolver needs dt to be specified when adding a dynamics")
    integrator = kwargs.get("integrator", default_integrator)
    if integrator is None:
        integrator = integrators.ImplicitTrapezoid
    if not (inspect.isclass(integrator) and issubclass(integrator, integrators.SingleStepIntegrator)):
        raise ValueError("The integrator has been defined, but is not a subclass of SingleStepIntegrator")
    dt = kwargs["dt"]
    if isinstance(dt, (str, Symbol)):
        raise ValueError("dt as a variable makes the defects bilinear: not an affine problem")
    states, inputs = dynamics.state_variables(), dynamics.input_names()
    involved = [n for n in states + inputs if n != dynamics.lhs.time_name]
    horizons = [len(ms._knot_names(n)) for n in involved]
    n = min((h for h in horizons if h > 1), default=0)
    if n < 2 or any(len(ms._knot_names(s)) != n for s in states):
        raise ValueError("The state variables of a dynamics need a horizon of the same length, larger than one")
    if "max_steps" in kwargs:
        n = min(n, int(kwargs["max_steps"]))
    problem = ms.get_problem()
    base = name if name is not None else "dot(" + ", ".join(states) + ")"
    # initial conditions: x0 is a {state: value} dict, or one value for a single state
    if x0 is not None:
        if not isinstance(x0, dict):
            if len(states) != 1:
                raise ValueError("x0 without names needs a dynamics with one state")
            x0 = {states[0]: x0}
        for j, (key, value) in enumerate(x0.items()):
            state = getattr(key, "flat_name", key)
            label = (x0_name if x0_name is not None else state + "[0]") + "{" + str(j) + "}"
            problem.add_expression(mode, _knot_symbol(ms, state, 0) == value, name=label)

    def knot_values(k):
        return {v: _knot_symbol(ms, v, k) for v in involved}

    def rhs(variables, time):
        return dynamics.evaluate(variables, time)
    for i in range(n - 1):
        here, there = knot_values(i), knot_values(i + 1)
        integrated = integrators.step(integrator, rhs, here, there, dt, t0 + i * dt)
        for j, state in enumerate(states):
            problem.add_expression(mode, there[state] == integrated[state], name=base + "[" + str(i + 1) + "]{" + str(j) + "}")


def add_expression_to_horizon(ms, expression, mode=None, apply_to_first_elements=False, name=None, **kwargs):
    mode = ExpressionType.subject_to if mode is None else mode
    leaves = sorted(expression.leaves())
    known = [leaf for leaf in leaves if leaf in ms._names]
    lengths = [ms._names[leaf][0] for leaf in known if ms._names[leaf][0] > 1]
    if not lengths:
        raise ValueError("The expression does not involve any time-varying variable")
    n = min(lengths)
    if "max_steps" in kwargs:
        n = min(n, int(kwargs["max_steps"]))
    problem = ms.get_problem()
    base = name if name is not None else "expression"
    for i in range(0 if apply_to_first_elements else 1, n):
        mapping = {leaf: ms._names[leaf][1][min(i, len(ms._names[leaf][1]) - 1)] for leaf in known}
        problem.add_expression(mode, expression.renamed(mapping), name=base + "[" + str(i) + "]", **{k: v for k, v in kwargs.items() if k != "max_steps"})
