from .variables import (  # noqa: F401
    ContactPointDescriptor, ContactPointState, ContactPointStateDerivative, FeetContactPointDescriptors, FeetContactPoints,
    FloatingBaseSystem, FloatingBaseSystemState, FootContactState, FreeFloatingObject, FreeFloatingObjectState,
    FreeFloatingObjectStateDerivative, HumanoidState, KinematicTree, KinematicTreeState, KinematicTreeStateDerivative,
)
from .variables import FeetContactPhasesDescriptor, FootContactPhaseDescriptor  # noqa: F401,E402
from .transforms import SE3, SO3  # noqa: F401,E402
from .interpolators import (  # noqa: F401,E402
    feet_contact_points_interpolator, floating_base_system_state_interpolator, foot_contact_state_interpolator,
    free_floating_object_state_interpolator, humanoid_state_interpolator, kinematic_tree_state_interpolator, linear_interpolator,
    quaternion_slerp, transform_interpolator,
)
