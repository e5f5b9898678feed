"""hippopt_amd — MI355X-native multiple-shooting NLP-callback engine behind hippopt's
solver boundary."""

# the reference's planner-level names (hippopt/__init__.py): `import hippopt_amd as hp` keeps planner.py:65-80 reading the same
from .base import (  # noqa: F401,E402
    ExpressionType, MultipleShootingSolver, OptimalControlProblem, OptimizationObject, OptimizationProblem, OptimizationSolver, Output, OverridableParameter,
    OverridableVariable, Parameter, StorageType, TimeExpansion, TypedProblemError, Variable, default_composite_field,
    default_storage_field, time_varying_metadata,
)


def __getattr__(name):   # (the solver plugin needs numpy / ctypes / the library only when it is used)
    if name in ("HipNlpSolver", "HipFailure", "InitialGuessFailure"):
        from . import hipnlp_solver
        return getattr(hipnlp_solver, name)
    raise AttributeError(name)
