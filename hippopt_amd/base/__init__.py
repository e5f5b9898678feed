from .optimization_object import (  # noqa: F401
    CompositeType, OptimizationObject, OverridableParameter, OverridableVariable, Parameter, StorageType,
    TimeExpansion, Variable, default_composite_field, default_storage_field, default_storage_metadata, time_varying_metadata,
)
from .horizon import extend_structure_to_horizon, flattened_names  # noqa: F401
from .problem import ExpressionType, Output, ProblemNotSolvedException  # noqa: F401
from . import opti_callback  # noqa: F401,E402
from .optimal_control import (  # noqa: F401,E402
    MultipleShootingSolver, OptimalControlProblem, OptimalControlProblemInstance, OptimizationSolver, TypedProblemError,
)
from .optimization_problem import OptimizationProblem, OptimizationProblemInstance  # noqa: F401,E402
