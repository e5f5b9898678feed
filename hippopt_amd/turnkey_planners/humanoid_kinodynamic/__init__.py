from .planner import Planner  # noqa: F401
from .settings import Settings  # noqa: F401
from .variables import (  # noqa: F401
    ContactReferences, ExtendedContactPoint, ExtendedHumanoid, ExtendedHumanoidState, FeetContactPointsExtended,
    FeetReferences, FootReferences, References, Variables,
)
