from .variables import (  # noqa: F401
    ContactPointDescriptor, ContactPointState, ContactPointStateDerivative, FeetContactPointDescriptors, FeetContactPoints,
    FloatingBaseSystem, FloatingBaseSystemState, FootContactState, FreeFloatingObject, FreeFloatingObjectState,
    FreeFloatingObjectStateDerivative, HumanoidState, KinematicTree, KinematicTreeState, KinematicTreeStateDerivative,
)
