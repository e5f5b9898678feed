"""Inert stand-in for the `casadi` module: just enough names for the reference's
pure-structure code (OptimizationObject flatten/scan, horizon expansion) to import.
Used ONLY by tools/gen_structure_fixtures.py inside the build container; never shipped
to the GPU box and never imported by the product or the tests."""
inf = float("inf")
OP_LE = OP_LT = OP_EQ = 0


class MX:  # noqa: D101
    def __init__(self, *_, **__):
        pass


class DM:  # noqa: D101
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

    def solver(self, *_, **__):
        pass


class OptiCallback:  # noqa: D101
    def __init__(self, *_, **__):
        pass
