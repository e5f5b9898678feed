"""Kinematic-tree model handed to the engine (include/hipnlp.h: hipnlp_robot_model).

The reference obtains its model by letting adam-robotics parse the ergoCub URDF
(turnkey_planners/humanoid_kinodynamic/planner.py:43-50).  Neither adam nor the URDF is
available to this build, so the default model is a *synthetic* 23-DoF humanoid with the
ergoCub topology and joint order (main_periodic_step.py:24-48): pelvis root, 3-DoF torso
chain ending in the chest, two 4-DoF arms off the chest, two 6-DoF legs off the pelvis ending
in the soles.  Dimensions/masses are plausible, not ergoCub's.  Every joint frame carries a
small seeded rotation so no Jacobian entry is accidentally (structurally) zero.

The arrays follow adam's conventions: parent_T_child(s) =
[R_fix Rot(axis, s), o_fix]; inertials in the link frame.
"""
import dataclasses

import numpy as np

from . import _abi

JOINT_NAMES = [
    "torso_pitch", "torso_roll", "torso_yaw",
    "l_shoulder_pitch", "l_shoulder_roll", "l_shoulder_yaw", "l_elbow",
    "r_shoulder_pitch", "r_shoulder_roll", "r_shoulder_yaw", "r_elbow",
    "l_hip_pitch", "l_hip_roll", "l_hip_yaw", "l_knee", "l_ankle_pitch", "l_ankle_roll",
    "r_hip_pitch", "r_hip_roll", "r_hip_yaw", "r_knee", "r_ankle_pitch", "r_ankle_roll",
]
FRAME_NAMES = ["l_sole", "r_sole", "chest"]


def rot_from_rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def rot_axis_angle(axis, q):
    a = np.asarray(axis, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.cos(q) * (np.eye(3) - np.outer(a, a)) + np.sin(q) * K + np.outer(a, a)


def rot_from_quat_xyzw(q):
    v, w = np.asarray(q[:3], float), float(q[3])
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + 2 * w * K + 2 * K @ K


@dataclasses.dataclass
class RobotModel:
    parent: np.ndarray       # [NJ] parent link of joint j (child link is j+1)
    R_fix: np.ndarray        # [NJ,3,3]
    o_fix: np.ndarray        # [NJ,3]
    axis: np.ndarray         # [NJ,3]
    mass: np.ndarray         # [NL]
    com: np.ndarray          # [NL,3]
    inertia: np.ndarray      # [NL,3,3]
    frame_link: np.ndarray   # [3]
    frame_R: np.ndarray      # [3,3,3]
    frame_o: np.ndarray      # [3,3]
    joint_names: list = dataclasses.field(default_factory=lambda: list(JOINT_NAMES))
    min_joint_positions: np.ndarray = None
    max_joint_positions: np.ndarray = None
    # every named frame of the source description: name -> (moving link it is rigidly attached to, link_R_frame, link_o_frame);
    # what a frame NAME of the reference's settings (e.g. left_hand_frame_name) resolves to.  None: names unknown.
    named_frames: dict = None

    def resolve_frame(self, name) -> tuple:
        if not self.named_frames or name not in self.named_frames:
            raise ValueError("frame '%s' is not known to this model" % name)
        link, R, o = self.named_frames[name]
        return int(link), np.array(R, float), np.array(o, float)

    @property
    def NDoF(self):  # noqa: N802  (adam KinDynComputations.NDoF, used at variables.py:319)
        return int(self.parent.shape[0])

    def get_total_mass(self):  # adam KinDynComputations.get_total_mass, planner.py:50
        return float(np.sum(self.mass))

    def to_c(self) -> _abi.RobotModelC:
        m = _abi.RobotModelC()
        for j in range(_abi.NJ):
            m.parent[j] = int(self.parent[j])
            for i in range(9):
                m.R_fix[j][i] = float(self.R_fix[j].reshape(9)[i])
            for i in range(3):
                m.o_fix[j][i] = float(self.o_fix[j][i])
                m.axis[j][i] = float(self.axis[j][i])
        for l in range(_abi.NL):
            m.mass[l] = float(self.mass[l])
            for i in range(3):
                m.com[l][i] = float(self.com[l][i])
            for i in range(9):
                m.inertia[l][i] = float(self.inertia[l].reshape(9)[i])
        for f in range(3):
            m.frame_link[f] = int(self.frame_link[f])
            for i in range(9):
                m.frame_R[f][i] = float(self.frame_R[f].reshape(9)[i])
            for i in range(3):
                m.frame_o[f][i] = float(self.frame_o[f][i])
        return m

    # ---- host-side numpy kinematics: only used to build synthetic inputs / initial guesses -------
    def link_poses(self, pb, quat_xyzw, s):
        q = np.asarray(quat_xyzw, float)
        q = q / np.linalg.norm(q)
        R = [None] * (self.NDoF + 1)
        o = [None] * (self.NDoF + 1)
        R[0], o[0] = rot_from_quat_xyzw(q), np.asarray(pb, float)
        for j in self.evaluation_order():
            par = int(self.parent[j])
            R[j + 1] = R[par] @ self.R_fix[j] @ rot_axis_angle(self.axis[j], s[j])
            o[j + 1] = o[par] + R[par] @ self.o_fix[j]
        return R, o

    def evaluation_order(self):
        """joints ordered so that each comes after the joint moving its parent link (the numbering is joints_name_list order,
        which need not follow the tree)"""
        done, order = {0}, []
        while len(order) < self.NDoF:
            progressed = False
            for j in range(self.NDoF):
                if (j + 1) not in done and int(self.parent[j]) in done:
                    done.add(j + 1)
                    order.append(j)
                    progressed = True
            if not progressed:
                raise ValueError("the parent links do not form a tree rooted at link 0")
        return order

    def frame_pose(self, frame, pb, quat_xyzw, s):
        R, o = self.link_poses(pb, quat_xyzw, s)
        l = int(self.frame_link[frame])
        return R[l] @ self.frame_R[frame], o[l] + R[l] @ self.frame_o[frame]

    def com_position(self, pb, quat_xyzw, s):
        R, o = self.link_poses(pb, quat_xyzw, s)
        acc = np.zeros(3)
        for l in range(self.NDoF + 1):
            acc += self.mass[l] * (o[l] + R[l] @ self.com[l])
        return acc / self.get_total_mass()


def synthetic_ergocub(seed: int = 0) -> RobotModel:
    """Seedable synthetic humanoid with ergoCub's topology (see module docstring)."""
    rng = np.random.RandomState(seed)
    nj, nl = _abi.NJ, _abi.NL
    parent = np.zeros(nj, np.int32)
    o_fix = np.zeros((nj, 3))
    axis = np.zeros((nj, 3))
    ax = {"x": [1, 0, 0], "y": [0, 1, 0], "z": [0, 0, 1]}

    def put(j, par, o, a):
        parent[j] = par
        o_fix[j] = o
        axis[j] = ax[a]

    # torso chain off the root (link 0); child link of joint j is j+1
    put(0, 0, [0.0, 0.0, 0.10], "y")
    put(1, 1, [0.0, 0.0, 0.02], "x")
    put(2, 2, [0.0, 0.0, 0.02], "z")   # -> link 3 = chest
    for side, j0 in ((+1, 3), (-1, 7)):  # arms off the chest
        put(j0 + 0, 3, [0.0, side * 0.11, 0.20], "y")
        put(j0 + 1, j0 + 1, [0.0, side * 0.03, 0.0], "x")
        put(j0 + 2, j0 + 2, [0.0, 0.0, -0.08], "z")
        put(j0 + 3, j0 + 3, [0.0, 0.0, -0.14], "y")
    for side, j0 in ((+1, 11), (-1, 17)):  # legs off the root
        put(j0 + 0, 0, [0.0, side * 0.075, -0.05], "y")
        put(j0 + 1, j0 + 1, [0.0, side * 0.02, 0.0], "x")
        put(j0 + 2, j0 + 2, [0.0, 0.0, -0.06], "z")
        put(j0 + 3, j0 + 3, [0.0, 0.0, -0.20], "y")
        put(j0 + 4, j0 + 4, [0.0, 0.0, -0.23], "y")
        put(j0 + 5, j0 + 5, [0.0, 0.0, -0.03], "x")
    R_fix = np.zeros((nj, 3, 3))
    for j in range(nj):
        R_fix[j] = rot_from_rpy(*(0.04 * rng.uniform(-1, 1, 3)))
        a = axis[j] + 0.03 * rng.uniform(-1, 1, 3)
        axis[j] = a / np.linalg.norm(a)
    link_mass = np.array([
        8.0, 1.5, 1.5, 10.0,
        1.5, 1.0, 1.0, 1.2, 1.5, 1.0, 1.0, 1.2,
        2.5, 1.5, 2.0, 2.5, 0.8, 1.5, 2.5, 1.5, 2.0, 2.5, 0.8, 1.5])
    assert link_mass.shape[0] == nl
    com = 0.03 * rng.uniform(-1, 1, (nl, 3))
    com[:, 2] -= 0.03
    com[0] = [0.0, 0.0, 0.02]
    com[3] = [0.0, 0.0, 0.12]
    inertia = np.zeros((nl, 3, 3))
    for l in range(nl):
        d = link_mass[l] * (0.04 + 0.04 * rng.uniform(0, 1, 3)) ** 2
        Q = rot_from_rpy(*(0.5 * rng.uniform(-1, 1, 3)))
        inertia[l] = Q @ np.diag(d) @ Q.T
        inertia[l] = 0.5 * (inertia[l] + inertia[l].T)
    frame_link = np.array([17, 23, 3], np.int32)  # l_ankle_roll child, r_ankle_roll child, chest
    frame_R = np.stack([rot_from_rpy(*(0.02 * rng.uniform(-1, 1, 3))) for _ in range(3)])
    frame_o = np.array([[0.03, 0.0, -0.06], [0.03, 0.0, -0.06], [0.0, 0.0, 0.15]])
    lim = np.array([0.6, 0.4, 0.6] + [1.6, 1.4, 1.0, 1.6] * 2 + [1.4, 0.6, 0.8, 1.6, 0.7, 0.4] * 2)
    named = {n: (int(frame_link[f]), frame_R[f], frame_o[f]) for f, n in enumerate(("l_sole", "r_sole", "chest"))}
    named["l_hand_palm"] = (7, rot_from_rpy(0.1, -0.2, 0.3), np.array([0.02, 0.01, -0.05]))    # ends of the two arm chains
    named["r_hand_palm"] = (11, rot_from_rpy(0.0, 0.1, 0.0), np.array([0.0, 0.0, -0.04]))
    return RobotModel(parent, R_fix, o_fix, axis, link_mass, com, inertia, frame_link, frame_R, frame_o,
                      min_joint_positions=-lim, max_joint_positions=lim, named_frames=named)
