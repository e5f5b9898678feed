"""Minimal FUNCTIONAL stand-in for the part of the CasADi Python API that hippopt's kinodynamic path touches.

PURPOSE: CasADi is not installable in the build container.  To pin the *assembly* of the NLP (which constraint
goes where, names, knot ranges, formulas as coded in robot_planning/expressions) the reference's OWN Python is
executed on top of this stand-in by tools/gen_planner_fixtures.py, and the resulting g / f / jac are committed as
golden vectors.  This is NOT CasADi: its numerics, AD, sparsity detection and Opti canonicalisation are restated
from CasADi's published behaviour (and labelled "unpinned" in DESIGN.md).  Never shipped, never imported by the
product or by the tests.

Everything is a dense 2-D matrix expression (MX) over numpy; evaluation and forward-mode derivatives are
vectorised over all directions at once.
"""
import numpy as np

inf = float("inf")
OP_EQ, OP_LE, OP_LT = 101, 102, 103
_UNARY = {
    "tanh": (np.tanh, lambda x, y: 1.0 - y * y),
    "sin": (np.sin, lambda x, y: np.cos(x)),
    "cos": (np.cos, lambda x, y: -np.sin(x)),
    "exp": (np.exp, lambda x, y: y),
    "sqrt": (np.sqrt, lambda x, y: 0.5 / y),
}


def _as2d(v):
    a = np.asarray(v, dtype=float)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        a = a.reshape(-1, 1)
    return a


class MX:
    __array_ufunc__ = None
    __array_priority__ = 1000
    _count = 0

    def __init__(self, *args):
        self.op, self.args, self.aux = "const", (), None
        if len(args) == 0:
            self.value = np.zeros((0, 0))
        elif len(args) == 1:
            a = args[0]
            if isinstance(a, MX):
                self.op, self.args, self.aux = a.op, a.args, a.aux
                self.value = getattr(a, "value", None)
                self._shape = a._shape
                self._name = getattr(a, "_name", None)
                self._id = a._id
                return
            self.value = _as2d(a)
        else:
            self.value = np.zeros((int(args[0]), int(args[1])))
        self._shape = self.value.shape
        MX._count += 1
        self._id = MX._count

    # ---- construction ----------------------------------------------------------------------------
    @staticmethod
    def _node(op, args, shape, aux=None):
        m = MX.__new__(MX)
        m.op, m.args, m.aux, m._shape = op, tuple(args), aux, tuple(shape)
        MX._count += 1
        m._id = MX._count
        return m

    @staticmethod
    def sym(name, n=1, m=1):
        r = MX._node("sym", (), (int(n), int(m)))
        r._name = name
        return r

    @staticmethod
    def zeros(n=1, m=1):
        return MX(np.zeros((n, m)))

    @staticmethod
    def ones(n=1, m=1):
        return MX(np.ones((n, m)))

    @staticmethod
    def eye(n):
        return MX(np.eye(n))

    # ---- introspection ---------------------------------------------------------------------------
    @property
    def shape(self):
        return self._shape

    def size1(self):
        return self._shape[0]

    def size2(self):
        return self._shape[1]

    def rows(self):
        return self._shape[0]

    def columns(self):
        return self._shape[1]

    def numel(self):
        return self._shape[0] * self._shape[1]

    def is_scalar(self):
        return self._shape == (1, 1)

    def is_vector(self):
        return 1 in self._shape

    def is_symbolic(self):
        return self.op == "sym"

    def is_constant(self):
        return self.op == "const"

    def name(self):
        if self.op != "sym":
            raise RuntimeError("name() of a non-symbolic MX")
        return self._name

    def is_op(self, code):
        return self.op == "cmp" and self.aux == code

    def dep(self, i=0):
        return self.args[i]

    def n_dep(self):
        return len(self.args)

    def __hash__(self):
        return self._id

    def __bool__(self):
        raise TypeError("truth value of a symbolic MX")

    def __repr__(self):
        return "MX(%s%s)" % (self.op, self._shape) if self.op != "sym" else "MX(%s)" % self._name

    def __str__(self):
        return self.__repr__()

    # ---- arithmetic ------------------------------------------------------------------------------
    @staticmethod
    def _wrap(v):
        if isinstance(v, MX):
            return v
        if hasattr(v, "xyzw"):  # a liecasadi Quaternion handed to cs.Function (expressions/quaternion.py:69-81): its coefficients
            return MX._wrap(v.xyzw)
        return MX(v)

    @staticmethod
    def _bshape(a, b):
        if a._shape == b._shape:
            return a._shape
        if a._shape == (1, 1):
            return b._shape
        if b._shape == (1, 1):
            return a._shape
        # CasADi >= 3.6: a column (n x 1) combined with a matrix (n x m) is repeated over the columns
        if a._shape[0] == b._shape[0] and (a._shape[1] == 1 or b._shape[1] == 1):
            return (a._shape[0], max(a._shape[1], b._shape[1]))
        raise RuntimeError("dimension mismatch %s vs %s" % (a._shape, b._shape))

    def _bin(self, other, op, swap=False):
        a, b = MX._wrap(self), MX._wrap(other)
        if swap:
            a, b = b, a
        if a.op == "const" and b.op == "const":
            f = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide}[op]
            return DM(f(a.value, b.value))
        return MX._node(op, (a, b), MX._bshape(a, b))

    def __add__(self, o): return self._bin(o, "add")
    def __radd__(self, o): return self._bin(o, "add", True)
    def __sub__(self, o): return self._bin(o, "sub")
    def __rsub__(self, o): return self._bin(o, "sub", True)
    def __mul__(self, o): return self._bin(o, "mul")
    def __rmul__(self, o): return self._bin(o, "mul", True)
    def __truediv__(self, o): return self._bin(o, "div")
    def __rtruediv__(self, o): return self._bin(o, "div", True)

    def __neg__(self):
        if self.op == "const":
            return DM(-self.value)
        return MX._node("neg", (self,), self._shape)

    def __pow__(self, e):
        return constpow(self, e)

    def __matmul__(self, o):
        return mtimes(self, o)

    def __rmatmul__(self, o):
        return mtimes(o, self)

    @property
    def T(self):
        if self.op == "const":
            return DM(self.value.T)
        return MX._node("transpose", (self,), (self._shape[1], self._shape[0]))

    def _cmp(self, o, code, swap=False):
        a, b = MX._wrap(self), MX._wrap(o)
        if swap:
            a, b = b, a
        return MX._node("cmp", (a, b), MX._bshape(a, b), code)

    def __eq__(self, o): return self._cmp(o, OP_EQ)  # noqa: E704
    def __le__(self, o): return self._cmp(o, OP_LE)  # noqa: E704
    def __lt__(self, o): return self._cmp(o, OP_LT)  # noqa: E704
    def __ge__(self, o): return self._cmp(o, OP_LE, True)  # a >= b  is  b <= a
    def __gt__(self, o): return self._cmp(o, OP_LT, True)

    def __getitem__(self, idx):
        n, m = self._shape
        if not isinstance(idx, tuple):
            if m == 1:
                idx = (idx, 0)
            elif n == 1:
                idx = (0, idx)
            else:  # linear (column-major) indexing of a matrix is not needed by the path
                raise NotImplementedError("linear indexing of a matrix")
        r, c = idx

        def norm(i, size):
            if isinstance(i, slice):
                return list(range(*i.indices(size)))
            if isinstance(i, (list, np.ndarray)):
                return [int(v) for v in i]
            i = int(i)
            return [i + size if i < 0 else i]
        rr, cc = norm(r, n), norm(c, m)
        if self.op == "const":
            return DM(self.value[np.ix_(rr, cc)])
        return MX._node("index", (self,), (len(rr), len(cc)), (tuple(rr), tuple(cc)))

    def nz(self):
        return self


class DM(MX):
    def __init__(self, *args):
        if len(args) == 1 and isinstance(args[0], MX):
            if args[0].op != "const":
                raise TypeError("DM from a symbolic expression")
            MX.__init__(self, args[0].value)
        else:
            MX.__init__(self, *args)

    def full(self):
        return np.array(self.value)

    def __float__(self):
        return float(self.value.reshape(-1)[0])

    @staticmethod
    def zeros(n=1, m=1):
        return DM(np.zeros((n, m)))

    @staticmethod
    def ones(n=1, m=1):
        return DM(np.ones((n, m)))

    @staticmethod
    def eye(n):
        return DM(np.eye(n))

    def __repr__(self):
        return "DM(%s)" % (self.value.tolist(),)


SX = MX


def DM_eye(n):
    return DM.eye(n)


# ---- free functions ------------------------------------------------------------------------------
def mtimes(a, b):
    a, b = MX._wrap(a), MX._wrap(b)
    if a._shape == (1, 1) or b._shape == (1, 1):
        return a * b
    if a._shape[1] != b._shape[0]:
        raise RuntimeError("mtimes dimension mismatch %s x %s" % (a._shape, b._shape))
    if a.op == "const" and b.op == "const":
        return DM(a.value @ b.value)
    return MX._node("matmul", (a, b), (a._shape[0], b._shape[1]))


def vertcat(*xs):
    xs = [MX._wrap(x) for x in xs if MX._wrap(x).numel() > 0 or True]
    xs = [x for x in xs if x._shape[0] > 0]
    if not xs:
        return MX(np.zeros((0, 1)))
    cols = xs[0]._shape[1]
    if all(x.op == "const" for x in xs):
        return DM(np.vstack([x.value for x in xs]))
    return MX._node("vertcat", xs, (sum(x._shape[0] for x in xs), cols))


def horzcat(*xs):
    xs = [MX._wrap(x) for x in xs]
    if all(x.op == "const" for x in xs):
        return DM(np.hstack([x.value for x in xs]))
    return MX._node("horzcat", xs, (xs[0]._shape[0], sum(x._shape[1] for x in xs)))


def veccat(*xs):
    return vertcat(*[MX._wrap(x) for x in xs])


def _unary(name, x):
    x = MX._wrap(x)
    if x.op == "const":
        return DM(_UNARY[name][0](x.value))
    return MX._node("unary", (x,), x._shape, name)


def tanh(x): return _unary("tanh", x)  # noqa: E704
def sin(x): return _unary("sin", x)  # noqa: E704
def cos(x): return _unary("cos", x)  # noqa: E704
def exp(x): return _unary("exp", x)  # noqa: E704
def sqrt(x): return _unary("sqrt", x)  # noqa: E704


def constpow(x, e):
    x = MX._wrap(x)
    e = float(e.value.reshape(-1)[0]) if isinstance(e, MX) else float(e)
    if x.op == "const":
        return DM(np.power(x.value, e))
    return MX._node("pow", (x,), x._shape, e)


def sumsqr(x):
    x = MX._wrap(x)
    return MX._node("sumall", (x * x,), (1, 1)) if x.op != "const" else DM(np.sum(x.value ** 2))


def sum1(x):
    x = MX._wrap(x)
    return mtimes(DM.ones(1, x._shape[0]), x)


def sum2(x):
    x = MX._wrap(x)
    return mtimes(x, DM.ones(x._shape[1], 1))


def norm_2(x):
    return sqrt(sumsqr(x))


def dot(a, b):
    return sumsqr_like(a, b)


def sumsqr_like(a, b):
    return MX._node("sumall", (MX._wrap(a) * MX._wrap(b),), (1, 1))


def cross(a, b):
    a, b = MX._wrap(a), MX._wrap(b)
    return vertcat(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def skew(v):
    v = MX._wrap(v)
    z = DM(0.0)
    return vertcat(horzcat(z, -v[2], v[1]), horzcat(v[2], z, -v[0]), horzcat(-v[1], v[0], z))


def mpower(a, n):
    r = a
    for _ in range(int(n) - 1):
        r = mtimes(r, a)
    return r


def diag(x):
    x = MX._wrap(x)
    n, m = x._shape
    if n == 1 or m == 1:  # vector -> diagonal matrix
        k = max(n, m)
        v = x if m == 1 else x.T
        if v.op == "const":
            return DM(np.diag(v.value.reshape(-1)))
        E = np.zeros((k * k, k))
        for i in range(k):
            E[i * k + i, i] = 1.0
        return MX._node("reshape", (mtimes(DM(E), v),), (k, k))
    if x.op == "const":
        return DM(np.diag(x.value).reshape(-1, 1))
    return vertcat(*[x[i, i] for i in range(n)])


def trace(x):
    x = MX._wrap(x)
    r = x[0, 0]
    for i in range(1, x._shape[0]):
        r = r + x[i, i]
    return r


def transpose(x):
    return MX._wrap(x).T


def Opti_bounded(lb, expr, ub):
    # CasADi: lb <= expr <= ub  ==  le(le(lb, expr), ub)
    inner = MX._wrap(lb)._cmp(expr, OP_LE)
    return inner._cmp(ub, OP_LE)


# ---- graph utilities -----------------------------------------------------------------------------
def _topo(roots):
    order, seen = [], set()
    stack = [(r, False) for r in roots]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n)
            continue
        if n._id in seen:
            continue
        seen.add(n._id)
        stack.append((n, True))
        for a in n.args:
            if a._id not in seen:
                stack.append((a, False))
    return order


def symvar(expr):
    out = []
    for n in _topo([MX._wrap(expr)]):
        if n.op == "sym":
            out.append(n)
    out.sort(key=lambda s: s._id)
    return out


def _rebuild(n, args):
    if n.op in ("add", "sub", "mul", "div"):
        return args[0]._bin(args[1], n.op)
    if n.op == "neg":
        return -args[0]
    if n.op == "transpose":
        return args[0].T
    if n.op == "matmul":
        return mtimes(args[0], args[1])
    if n.op == "index":
        return args[0][list(n.aux[0]), list(n.aux[1])]
    if n.op == "vertcat":
        return vertcat(*args)
    if n.op == "horzcat":
        return horzcat(*args)
    if n.op == "unary":
        return _unary(n.aux, args[0])
    if n.op == "pow":
        return constpow(args[0], n.aux)
    if n.op == "sumall":
        a = args[0]
        return DM(np.sum(a.value)) if a.op == "const" else MX._node("sumall", (a,), (1, 1))
    if n.op == "reshape":
        a = args[0]
        return DM(a.value.reshape(n._shape)) if a.op == "const" else MX._node("reshape", (a,), n._shape)
    if n.op == "cmp":
        return args[0]._cmp(args[1], n.aux)
    raise NotImplementedError(n.op)


def substitute(exprs, vs, vals):
    single = isinstance(exprs, MX)
    exprs_l = [exprs] if single else list(exprs)
    vs = [vs] if isinstance(vs, MX) else list(vs)
    vals = [vals] if isinstance(vals, MX) or not isinstance(vals, (list, tuple)) else list(vals)
    memo = {v._id: MX._wrap(val) for v, val in zip(vs, vals)}
    for n in _topo([MX._wrap(e) for e in exprs_l]):
        if n._id in memo:
            continue
        if n.op in ("sym", "const"):
            memo[n._id] = n
        else:
            new_args = [memo[a._id] for a in n.args]
            memo[n._id] = n if all(x is y for x, y in zip(new_args, n.args)) else _rebuild(n, new_args)
    out = [memo[MX._wrap(e)._id] for e in exprs_l]
    return out[0] if single else out


def jtimes(ex, arg, v):
    """Forward directional derivative of ex with respect to the symbol arg along v, as a new expression."""
    ex, arg, v = MX._wrap(ex), MX._wrap(arg), MX._wrap(v)
    tang = {}
    zero = lambda n: DM(np.zeros(n._shape))  # noqa: E731
    for n in _topo([ex]):
        if n._id == arg._id:
            tang[n._id] = v
        elif n.op in ("sym", "const"):
            tang[n._id] = zero(n)
        else:
            a = n.args
            t = [tang[x._id] for x in a]
            if n.op == "add":
                tang[n._id] = t[0] + t[1]
            elif n.op == "sub":
                tang[n._id] = t[0] - t[1]
            elif n.op == "mul":
                tang[n._id] = t[0] * a[1] + a[0] * t[1]
            elif n.op == "div":
                tang[n._id] = (t[0] - n * t[1]) / a[1]
            elif n.op == "neg":
                tang[n._id] = -t[0]
            elif n.op == "transpose":
                tang[n._id] = t[0].T
            elif n.op == "matmul":
                tang[n._id] = mtimes(t[0], a[1]) + mtimes(a[0], t[1])
            elif n.op == "index":
                tang[n._id] = t[0][list(n.aux[0]), list(n.aux[1])]
            elif n.op == "vertcat":
                tang[n._id] = vertcat(*t)
            elif n.op == "horzcat":
                tang[n._id] = horzcat(*t)
            elif n.op == "unary":
                d = {"tanh": lambda x, y: 1.0 - y * y, "sin": lambda x, y: cos(x), "cos": lambda x, y: -sin(x),
                     "exp": lambda x, y: y, "sqrt": lambda x, y: 0.5 / y}[n.aux](a[0], n)
                tang[n._id] = d * t[0]
            elif n.op == "pow":
                tang[n._id] = (n.aux * constpow(a[0], n.aux - 1.0)) * t[0]
            elif n.op == "sumall":
                tang[n._id] = MX._node("sumall", (t[0],), (1, 1)) if t[0].op != "const" else DM(np.sum(t[0].value))
            elif n.op == "reshape":
                tang[n._id] = MX._node("reshape", (t[0],), n._shape) if t[0].op != "const" else DM(t[0].value.reshape(n._shape))
            else:
                raise NotImplementedError(n.op)
    return tang[ex._id]


def gradient(ex, arg):
    arg = MX._wrap(arg)
    cols = []
    for i in range(arg._shape[0]):
        e = np.zeros(arg._shape)
        e[i, 0] = 1.0
        cols.append(jtimes(ex, arg, DM(e)))
    return vertcat(*cols)


def jacobian(ex, arg):
    arg = MX._wrap(arg)
    cols = []
    for i in range(arg._shape[0]):
        e = np.zeros(arg._shape)
        e[i, 0] = 1.0
        cols.append(jtimes(ex, arg, DM(e)))
    return horzcat(*cols)


# ---- numeric evaluation with vectorised forward mode ---------------------------------------------
def evaluate(roots, values, seeds=None, ndir=0):
    """values: {sym id: ndarray}; seeds: {sym id: ndarray [rows, cols, ndir]}.  Returns ([values], [tangents])."""
    val, tan = {}, {}
    want_t = ndir > 0
    for n in _topo([MX._wrap(r) for r in roots]):
        if n.op == "const":
            v = n.value
            t = None
        elif n.op == "sym":
            v = values[n._id]
            t = seeds.get(n._id) if want_t else None
        else:
            av = [val[a._id] for a in n.args]
            at = [tan[a._id] for a in n.args] if want_t else None
            t = None
            if n.op in ("add", "sub", "mul", "div"):
                a, b = av
                if a.shape != b.shape and a.shape != (1, 1) and b.shape != (1, 1):
                    a = np.broadcast_to(a, n._shape)
                    b = np.broadcast_to(b, n._shape)
                if n.op == "add":
                    v = a + b
                elif n.op == "sub":
                    v = a - b
                elif n.op == "mul":
                    v = a * b
                else:
                    v = a / b
                if want_t and (at[0] is not None or at[1] is not None):
                    def bt(x, ref):
                        if x is None:
                            return None
                        return np.broadcast_to(x, n._shape + (ndir,)) if x.shape[:2] != n._shape else x
                    ta, tb = bt(at[0], a), bt(at[1], b)
                    A = np.broadcast_to(a, n._shape)[..., None]
                    B = np.broadcast_to(b, n._shape)[..., None]
                    z = 0.0
                    if n.op == "add":
                        t = (ta if ta is not None else z) + (tb if tb is not None else z)
                    elif n.op == "sub":
                        t = (ta if ta is not None else z) - (tb if tb is not None else z)
                    elif n.op == "mul":
                        t = (ta * B if ta is not None else z) + (A * tb if tb is not None else z)
                    else:
                        t = ((ta if ta is not None else z) - v[..., None] * (tb if tb is not None else z)) / B
                    if np.isscalar(t):
                        t = None
            elif n.op == "neg":
                v = -av[0]
                t = -at[0] if want_t and at[0] is not None else None
            elif n.op == "transpose":
                v = av[0].T
                t = np.swapaxes(at[0], 0, 1) if want_t and at[0] is not None else None
            elif n.op == "matmul":
                v = av[0] @ av[1]
                if want_t and (at[0] is not None or at[1] is not None):
                    t = 0.0
                    if at[0] is not None:
                        t = t + np.einsum("ikd,kj->ijd", at[0], av[1])
                    if at[1] is not None:
                        t = t + np.einsum("ik,kjd->ijd", av[0], at[1])
            elif n.op == "index":
                ix = np.ix_(n.aux[0], n.aux[1])
                v = av[0][ix]
                t = at[0][ix] if want_t and at[0] is not None else None
            elif n.op in ("vertcat", "horzcat"):
                axis = 0 if n.op == "vertcat" else 1
                v = np.concatenate(av, axis=axis)
                if want_t and any(x is not None for x in at):
                    t = np.concatenate([x if x is not None else np.zeros(a.shape + (ndir,)) for x, a in zip(at, av)], axis=axis)
            elif n.op == "unary":
                f, d = _UNARY[n.aux]
                v = f(av[0])
                t = d(av[0], v)[..., None] * at[0] if want_t and at[0] is not None else None
            elif n.op == "pow":
                v = np.power(av[0], n.aux)
                t = (n.aux * np.power(av[0], n.aux - 1.0))[..., None] * at[0] if want_t and at[0] is not None else None
            elif n.op == "sumall":
                v = np.sum(av[0]).reshape(1, 1)
                t = np.sum(at[0], axis=(0, 1)).reshape(1, 1, ndir) if want_t and at[0] is not None else None
            elif n.op == "reshape":
                v = av[0].reshape(n._shape)
                t = at[0].reshape(n._shape + (ndir,)) if want_t and at[0] is not None else None
            elif n.op == "cmp":
                raise RuntimeError("cannot evaluate a comparison node")
            else:
                raise NotImplementedError(n.op)
        val[n._id] = v
        tan[n._id] = t
    rs = [MX._wrap(r) for r in roots]
    return [val[r._id] for r in rs], [tan[r._id] for r in rs]


def depends_on(roots):
    """Structural dependency (set of symbol ids per root ENTRY) with SX-like zero simplification is not attempted here;
    the fixture generator derives the pattern from numeric tangents at two random points instead."""
    raise NotImplementedError


# ---- Function ----------------------------------------------------------------------------------------
class Function:
    def __init__(self, name, ins, outs, names_in=None, names_out=None, opts=None):
        if isinstance(names_in, dict) and names_out is None:
            names_in, opts = None, names_in
        self._name = name
        self._ins = [MX._wrap(i) for i in ins]
        for i in self._ins:
            if i.op != "sym":
                raise RuntimeError("Function inputs must be symbolic")
        self._outs = [MX._wrap(o) for o in outs]
        self._names_in = list(names_in) if names_in is not None else ["i%d" % k for k in range(len(ins))]
        self._names_out = list(names_out) if names_out is not None else ["o%d" % k for k in range(len(outs))]

    def name(self):
        return self._name

    def name_in(self, i=None):
        return list(self._names_in) if i is None else self._names_in[i]

    def name_out(self, i=None):
        return list(self._names_out) if i is None else self._names_out[i]

    def n_in(self):
        return len(self._ins)

    def n_out(self):
        return len(self._outs)

    def numel_in(self, i=None):
        return sum(x.numel() for x in self._ins) if i is None else self._ins[self._names_in.index(i) if isinstance(i, str) else i].numel()

    def numel_out(self, i=None):
        return sum(x.numel() for x in self._outs) if i is None else self._outs[i].numel()

    def size_in(self, i):
        k = self._names_in.index(i) if isinstance(i, str) else i
        return self._ins[k]._shape

    def _apply(self, vals):
        vals = [MX._wrap(v) for v in vals]
        for k, (v, i) in enumerate(zip(vals, self._ins)):
            if v._shape != i._shape:
                if v._shape == (i._shape[1], i._shape[0]) and 1 in v._shape:
                    vals[k] = v.T
                else:
                    raise RuntimeError("%s: input %s expects shape %s, got %s" % (self._name, self._names_in[k], i._shape, v._shape))
        return substitute(self._outs, self._ins, vals)

    def __call__(self, *args, **kwargs):
        if kwargs:
            vals = []
            for nm, i in zip(self._names_in, self._ins):
                if nm not in kwargs:
                    raise RuntimeError("%s: missing input %s" % (self._name, nm))
                vals.append(kwargs[nm])
            outs = self._apply(vals)
            return {n: o for n, o in zip(self._names_out, outs)}
        if len(args) == 1 and isinstance(args[0], dict):
            return self.__call__(**args[0])
        outs = self._apply(args)
        return outs[0] if len(outs) == 1 else tuple(outs)


# ---- Opti (construction only; the fixture generator reads the recorded problem) ---------------------
class OptiSol:
    pass


class OptiAdvanced:
    pass


class OptiCallback:
    def __init__(self, *_, **__):
        pass


class Opti:
    def __init__(self, problem_type="nlp"):
        self.problem_type = problem_type
        self.variables, self.parameters, self.constraints = [], [], []
        self.objective = None
        self.initial, self.values = {}, {}

    def variable(self, n=1, m=1):
        v = MX.sym("opti_x_%d" % len(self.variables), n, m)
        self.variables.append(v)
        return v

    def parameter(self, n=1, m=1):
        p = MX.sym("opti_p_%d" % len(self.parameters), n, m)
        self.parameters.append(p)
        return p

    def subject_to(self, expr=None):
        if expr is None:
            self.constraints = []
            return
        self.constraints.append(expr)

    def minimize(self, f):
        self.objective = f

    def set_initial(self, var, val):
        self.initial[var._id] = _as2d(val.full() if isinstance(val, DM) else val)

    def set_value(self, par, val):
        self.values[par._id] = _as2d(val.full() if isinstance(val, DM) else val)

    def solver(self, *_, **__):
        pass

    def callback(self, *_, **__):
        pass

    def solve(self):
        raise RuntimeError("the CasADi stand-in cannot solve; it only records the problem")
