"""NUMERIC stand-in for the handful of casadi functions the reference's host-side utilities call on numbers
(robot_planning/utilities/interpolators.py: cs.acos, cs.fabs, cs.if_else, cs.DM.zeros) — numpy behind casadi's names, plus
the inert class names the rest of the reference needs at import time.  Used ONLY by tools/gen_interpolator_fixtures.py in the
build container (CasADi itself is not installed); never shipped, never imported by the product or the tests."""
import numpy as np

inf = float("inf")
OP_LE = OP_LT = OP_EQ = 0


class MX:  # noqa: D101  (import-time placeholder)
    def __init__(self, *_, **__):
        pass


class SX:  # noqa: D101
    pass


class Function:  # noqa: D101
    pass


class OptiSol:  # noqa: D101
    pass


class OptiAdvanced:  # noqa: D101
    pass


class Opti:  # noqa: D101
    def __init__(self, *_, **__):
        pass


class OptiCallback:  # noqa: D101
    def __init__(self, *_, **__):
        pass


class DM(np.ndarray):
    """Dense numeric matrix: a 2-D float ndarray (column vectors for 1-D input, like casadi.DM)."""

    def __new__(cls, data=0.0):
        a = np.array(data, dtype=float)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        return a.view(cls)

    @staticmethod
    def zeros(r, c=1):
        return DM(np.zeros((r, c)))

    @staticmethod
    def eye(n):
        return DM(np.eye(n))

    def full(self):
        return np.asarray(self)


def _a(x):
    return np.asarray(x, dtype=float)


def acos(x):
    return np.arccos(_a(x))


def fabs(x):
    return np.abs(_a(x))


def sin(x):
    return np.sin(_a(x))


def cos(x):
    return np.cos(_a(x))


def dot(a, b):
    return float(np.sum(_a(a) * _a(b)))


def if_else(cond, a, b):
    """casadi.if_else on numbers evaluates both branches and selects (a NaN in the unselected branch does not propagate)."""
    return _a(a) if bool(np.all(cond)) else _a(b)
