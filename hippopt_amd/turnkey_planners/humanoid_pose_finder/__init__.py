"""Static pose finder: host-side mirror of hippopt.turnkey_planners.humanoid_pose_finder.planner (Settings, References,
Variables, Planner) on the hipnlp_pose_* engine."""
from .planner import Planner, References, Settings, Variables  # noqa: F401
