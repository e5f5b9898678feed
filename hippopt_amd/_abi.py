"""ctypes mirror of include/hipnlp.h (POD descriptors).  Pure data layout, no compute."""
import ctypes as C

NJ, NL, NC, NXK, NPK, NXG, NPG = 23, 24, 8, 189, 79, 6, 326
NCOST_TERMS = 12
ABI_VERSION = 3
FLAG_DETECT_SIMPLE_BOUNDS = 1
FLAG_JAC_VARYING_FIRST = 2

EXPR_SKIP, EXPR_SUBJECT_TO, EXPR_MINIMIZE = 0, 1, 2
TERRAIN_PLANAR, TERRAIN_SMOOTH_STEPS, MAX_TERRAIN_STEPS = 0, 1, 4
FRAME_LEFT_SOLE, FRAME_RIGHT_SOLE, FRAME_CHEST = 0, 1, 2

OK, E_INVALID, E_NODEVICE, E_ALLOC, E_PARAMS, E_NUMERIC = 0, -1, -2, -3, -4, -5


class RobotModelC(C.Structure):
    _fields_ = [
        ("parent", C.c_int32 * NJ),
        ("R_fix", (C.c_double * 9) * NJ),
        ("o_fix", (C.c_double * 3) * NJ),
        ("axis", (C.c_double * 3) * NJ),
        ("mass", C.c_double * NL),
        ("com", (C.c_double * 3) * NL),
        ("inertia", (C.c_double * 9) * NL),
        ("frame_link", C.c_int32 * 3),
        ("frame_R", (C.c_double * 9) * 3),
        ("frame_o", (C.c_double * 3) * 3),
    ]


class TerrainStepC(C.Structure):
    _fields_ = [("length", C.c_double), ("width", C.c_double), ("height", C.c_double), ("position", C.c_double * 3),
                ("orientation", C.c_double), ("edge_sharpness", C.c_int32), ("side_sharpness", C.c_int32), ("top_normal", C.c_double * 3)]


class SettingsC(C.Structure):
    _fields_ = [
        ("horizon", C.c_int32),
        ("terrain", C.c_int32),
        ("final_state_type", C.c_int32),
        ("periodicity_type", C.c_int32),
        ("joint_reg_as_coded", C.c_int32),
        ("yaw_corner", (C.c_int32 * 3) * 2),
        ("final_state_weight", C.c_double),
        ("periodicity_weight", C.c_double),
        ("contacts_centroid_cost_multiplier", C.c_double),
        ("com_linear_velocity_cost_weights", C.c_double * 3),
        ("com_linear_velocity_cost_multiplier", C.c_double),
        ("desired_frame_quaternion_cost_multiplier", C.c_double),
        ("base_quaternion_cost_multiplier", C.c_double),
        ("base_quaternion_velocity_cost_multiplier", C.c_double),
        ("joint_regularization_cost_weights", C.c_double * NJ),
        ("joint_regularization_cost_multiplier", C.c_double),
        ("force_regularization_cost_multiplier", C.c_double),
        ("foot_yaw_regularization_cost_multiplier", C.c_double),
        ("swing_foot_height_cost_multiplier", C.c_double),
        ("contact_velocity_control_cost_multiplier", C.c_double),
        ("contact_force_control_cost_multiplier", C.c_double),
        ("n_terrain_steps", C.c_int32),
        ("reserved_", C.c_int32),
        ("terrain_steps", TerrainStepC * MAX_TERRAIN_STEPS),
    ]


class DescC(C.Structure):
    _fields_ = [
        ("settings", SettingsC),
        ("model", RobotModelC),
        ("batch", C.c_int32),
        ("knot_begin", C.c_int32),
        ("knot_end", C.c_int32),
        ("device", C.c_int32),
        ("abi_version", C.c_int32),
        ("flags", C.c_int32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.abi_version = ABI_VERSION


class DimsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "m", "nnz", "np", "nnz_knot", "m_knot",
        "shard_g_rows", "shard_nnz", "shard_grad", "shard_jac_off", "shard_grad_off", "m_full", "n_lifted")]


# ---- static pose finder (hipnlp_pose_*) ----------------------------------------------------------------------
POSE_NX, POSE_NP, POSE_NCOST_TERMS = 81, 202, 9


class PoseSettingsC(C.Structure):
    _fields_ = [
        ("terrain", C.c_int32),
        ("n_terrain_steps", C.c_int32),
        ("terrain_steps", TerrainStepC * MAX_TERRAIN_STEPS),
        ("com_position_type", C.c_int32),
        ("left_point_position_type", C.c_int32),
        ("right_point_position_type", C.c_int32),
        ("reserved_", C.c_int32),
        ("base_quaternion_cost_multiplier", C.c_double),
        ("desired_frame_quaternion_cost_multiplier", C.c_double),
        ("com_regularization_cost_multiplier", C.c_double),
        ("joint_regularization_cost_weights", C.c_double * NJ),
        ("joint_regularization_cost_multiplier", C.c_double),
        ("force_regularization_cost_multiplier", C.c_double),
        ("average_force_regularization_cost_multiplier", C.c_double),
        ("point_position_regularization_cost_multiplier", C.c_double),
        ("hand_type", C.c_int32 * 2),
        ("hand_frame_link", C.c_int32 * 2),
        ("hand_frame_R", (C.c_double * 9) * 2),
        ("hand_frame_o", (C.c_double * 3) * 2),
        ("hand_regularization_cost_multiplier", C.c_double * 2),
    ]


class PoseDescC(C.Structure):
    _fields_ = [("settings", PoseSettingsC), ("model", RobotModelC), ("batch", C.c_int32), ("device", C.c_int32),
                ("abi_version", C.c_int32), ("flags", C.c_int32)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.abi_version = ABI_VERSION


class PoseDimsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "m", "nnz", "np")]
