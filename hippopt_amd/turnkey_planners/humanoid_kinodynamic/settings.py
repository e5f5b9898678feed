"""Settings of the kinodynamic planner: the numeric mirror (hippopt_amd.kinodyn_settings.KinodynSettings) plus the
non-numeric fields of the reference's Settings (turnkey_planners/humanoid_kinodynamic/settings.py:12-147)."""
import dataclasses

from ... import robot_planning as hp_rp
from ...kinodyn_settings import KinodynSettings


@dataclasses.dataclass
class Settings(KinodynSettings):
    joints_name_list: list = None
    root_link: str = "root_link"
    contact_points: hp_rp.FeetContactPointDescriptors = None
    desired_frame_quaternion_cost_frame_name: str = "chest"
    use_opti_callback: bool = False                       # settings.py:75, :135-145
    acceptable_constraint_violation: float = 1e-3
    opti_callback_save_costs: bool = True
    opti_callback_save_constraint_multipliers: bool = True
    solver_options: dict = dataclasses.field(default_factory=dict)

    def __post_init__(self):
        KinodynSettings.__post_init__(self)
        if self.contact_points is None:
            self.contact_points = hp_rp.FeetContactPointDescriptors()
            self.contact_points.left = hp_rp.ContactPointDescriptor.rectangular_foot("l_sole", 0.232, 0.1, [0.116, 0.05, 0.0])
            self.contact_points.right = hp_rp.ContactPointDescriptor.rectangular_foot("r_sole", 0.232, 0.1, [0.116, 0.05, 0.0])

    @staticmethod
    def from_numeric(numeric: KinodynSettings, **kwargs):
        fields = {f.name: getattr(numeric, f.name) for f in dataclasses.fields(KinodynSettings)}
        fields.update(kwargs)
        return Settings(**fields)
