"""URDF -> hipnlp_robot_model: what stands where `adam.casadi.KinDynComputations(urdf, joints_name_list, root_link, gravity)`
stands in the reference (turnkey_planners/humanoid_kinodynamic/planner.py:43-50; URDF resolved at
main_single_step_flat_ground.py:17-19, joint list :22-46, root link :53, foot frames `l_sole` / `r_sole` :57-68, `chest` :95).

adam's conventions, restated (adam is not in the image):
* the joints of `joints_name_list` are the degrees of freedom, IN THAT ORDER (it fixes the layout of the joint vector s, and with it
  of x); every other joint of the URDF — `fixed` ones and movable ones that are not listed — is rigid at its zero position;
* parent_T_child(s_j) = T(origin xyz, origin rpy) * Rot(axis, s_j); rpy is roll-pitch-yaw about fixed axes (R = Rz Ry Rx);
* link inertials: mass, centre of mass and the inertia tensor about it, in the inertial frame `origin` places in the link frame;
* a frame is a link (usually massless, behind a fixed joint).

The engine's model has one link per degree of freedom plus the root.  Links behind rigid joints are therefore LUMPED into the moving
link that carries them (mass, first moment, inertia about the common centre of mass — exactly what summing over all links gives
adam for the CoM and the centroidal momentum), and a frame behind rigid joints becomes a fixed transform on its moving link.
The child link of listed joint j is engine link j + 1; parent[j] is the engine link that carries the joint's URDF parent link.
The joint order need not follow the tree.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import _abi
from .robot_model import FRAME_NAMES, RobotModel, rot_from_rpy


class UrdfError(ValueError):
    pass


def _vec(text, n=3, default=0.0):
    if text is None:
        return np.full(n, default, float)
    v = np.array([float(t) for t in text.split()], float)
    if v.size != n:
        raise UrdfError(f"expected {n} numbers, got '{text}'")
    return v


def _origin(elem):
    """(R, o) of an <origin xyz rpy> child (identity when absent)"""
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return np.eye(3), np.zeros(3)
    return rot_from_rpy(*_vec(o.get("rpy"))), _vec(o.get("xyz"))


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


class _Body:
    """mass properties accumulated in one moving link's frame: mass, first moment, inertia about the frame origin"""

    def __init__(self):
        self.m, self.h, self.I = 0.0, np.zeros(3), np.zeros((3, 3))

    def add(self, mass, com, inertia_com, R, o):
        """a link with (mass, com, inertia about its com) in its own frame, placed at (R, o) in this body's frame"""
        c = o + R @ com
        self.m += mass
        self.h += mass * c
        self.I += R @ inertia_com @ R.T - mass * _skew(c) @ _skew(c)

    def result(self):
        if self.m <= 0.0:
            return 0.0, np.zeros(3), np.zeros((3, 3))
        c = self.h / self.m
        Ic = self.I + self.m * _skew(c) @ _skew(c)
        return self.m, c, 0.5 * (Ic + Ic.T)


def parse_urdf(source):
    """{links: name -> (mass, com, inertia about com in the link frame), joints: [dict]} from a URDF path or string"""
    if isinstance(source, (bytes, str)) and not str(source).lstrip().startswith("<"):
        if not os.path.exists(source):
            raise UrdfError(f"URDF file not found: {source}")
        root = ET.parse(source).getroot()
    else:
        root = ET.fromstring(source)
    if root.tag != "robot":
        raise UrdfError("not a URDF: the root element is not <robot>")
    links, joints = {}, []
    for ln in root.findall("link"):
        name = ln.get("name")
        inert = ln.find("inertial")
        mass, com, I = 0.0, np.zeros(3), np.zeros((3, 3))
        if inert is not None:
            Ri, com = _origin(inert)
            me = inert.find("mass")
            mass = float(me.get("value")) if me is not None else 0.0
            ie = inert.find("inertia")
            if ie is not None:
                g = lambda k: float(ie.get(k, "0"))  # noqa: E731
                Ii = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
                I = Ri @ Ii @ Ri.T
        links[name] = (mass, com, I)
    for jn in root.findall("joint"):
        R, o = _origin(jn)
        ax = jn.find("axis")
        axis = _vec(ax.get("xyz")) if ax is not None else np.array([1.0, 0.0, 0.0])
        lim = jn.find("limit")
        joints.append({"name": jn.get("name"), "type": jn.get("type"), "parent": jn.find("parent").get("link"),
                       "child": jn.find("child").get("link"), "R": R, "o": o, "axis": axis,
                       "lower": float(lim.get("lower")) if lim is not None and lim.get("lower") is not None else -np.inf,
                       "upper": float(lim.get("upper")) if lim is not None and lim.get("upper") is not None else np.inf})
    return {"links": links, "joints": joints}


def load_urdf(source, joints_name_list, root_link="root_link", frames=tuple(FRAME_NAMES)) -> RobotModel:
    """The engine's robot model of a URDF (see the module docstring).  frames: the left sole, right sole and chest frame names
    (settings.contact_points.left/right foot_frame, settings.desired_frame_quaternion_cost_frame_name)."""
    u = parse_urdf(source)
    links, joints = u["links"], u["joints"]
    nj = len(joints_name_list)
    if nj != _abi.NJ:
        raise UrdfError(f"the engine is built for {_abi.NJ} degrees of freedom, joints_name_list has {nj}")
    if len(set(joints_name_list)) != nj:
        raise UrdfError("joints_name_list names a joint twice")
    if root_link not in links:
        raise UrdfError(f"root link '{root_link}' is not in the URDF")
    by_name = {j["name"]: j for j in joints}
    dof = {}
    for i, name in enumerate(joints_name_list):
        if name not in by_name:
            raise UrdfError(f"joint '{name}' of joints_name_list is not in the URDF")
        if by_name[name]["type"] not in ("revolute", "continuous"):
            raise UrdfError(f"joint '{name}' is {by_name[name]['type']}: the engine's degrees of freedom are revolute")
        dof[name] = i
    children = {}
    for j in joints:
        children.setdefault(j["parent"], []).append(j)
        if j["parent"] not in links or j["child"] not in links:
            raise UrdfError(f"joint '{j['name']}' connects a link the URDF does not define")
    # walk the tree from the root; every URDF link gets (engine link, R, o): its pose in that moving link's frame
    nl = nj + 1
    bodies = [_Body() for _ in range(nl)]
    placed = {root_link: (0, np.eye(3), np.zeros(3))}
    parent = np.full(nj, -1, np.int32)
    R_fix, o_fix, axis = np.zeros((nj, 3, 3)), np.zeros((nj, 3)), np.zeros((nj, 3))
    lo, hi = np.full(nj, -np.inf), np.full(nj, np.inf)
    stack = [root_link]
    while stack:
        name = stack.pop()
        e, R, o = placed[name]
        mass, com, I = links[name]
        if mass > 0.0:
            bodies[e].add(mass, com, I, R, o)
        for j in children.get(name, []):
            if j["child"] in placed:
                raise UrdfError(f"link '{j['child']}' has two parents: not a tree")
            Rj, oj = R @ j["R"], o + R @ j["o"]          # the joint frame in the moving link's frame
            if j["name"] in dof:
                i = dof[j["name"]]
                a = np.asarray(j["axis"], float)
                na = np.linalg.norm(a)
                if na == 0.0:
                    raise UrdfError(f"joint '{j['name']}' has a zero axis")
                parent[i], R_fix[i], o_fix[i], axis[i] = e, Rj, oj, a / na
                lo[i], hi[i] = j["lower"], j["upper"]
                placed[j["child"]] = (i + 1, np.eye(3), np.zeros(3))
            else:                                          # rigid at its zero position: fixed, or movable but not listed
                placed[j["child"]] = (e, Rj, oj)
            stack.append(j["child"])
    missing = [n for n in joints_name_list if parent[dof[n]] < 0]
    if missing:
        raise UrdfError(f"joints {missing} are not reachable from root link '{root_link}'")
    mass, com, inertia = np.zeros(nl), np.zeros((nl, 3)), np.zeros((nl, 3, 3))
    for l in range(nl):
        mass[l], com[l], inertia[l] = bodies[l].result()
    if len(frames) != 3:
        raise UrdfError("frames = (left sole, right sole, chest)")
    frame_link, frame_R, frame_o = np.zeros(3, np.int32), np.zeros((3, 3, 3)), np.zeros((3, 3))
    for f, fname in enumerate(frames):
        if fname not in placed:
            raise UrdfError(f"frame '{fname}' is not a link reachable from '{root_link}'")
        frame_link[f], frame_R[f], frame_o[f] = placed[fname]
    return RobotModel(parent, R_fix, o_fix, axis, mass, com, inertia, frame_link, frame_R, frame_o,
                      joint_names=list(joints_name_list), min_joint_positions=lo, max_joint_positions=hi,
                      named_frames={n: (int(v[0]), np.array(v[1]), np.array(v[2])) for n, v in placed.items()})


def rpy_from_rot(R):
    """roll-pitch-yaw (fixed axes, R = Rz Ry Rx) of a rotation matrix"""
    sy = -R[2, 0]
    if abs(sy) < 1.0 - 1e-12:
        return np.array([np.arctan2(R[2, 1], R[2, 2]), np.arcsin(sy), np.arctan2(R[1, 0], R[0, 0])])
    return np.array([0.0, np.pi / 2 * np.sign(sy), np.arctan2(-R[0, 1] * np.sign(sy), R[1, 1])])


def to_urdf(model: RobotModel, name="robot", root_link="root_link", frames=tuple(FRAME_NAMES), extra_frames=()) -> str:
    """A URDF of an engine model (one link per degree of freedom, the three frames as massless links behind fixed joints): the
    inverse of load_urdf up to rounding, used to exchange models with tools that read URDF and by the loader's round-trip tests.
    extra_frames: names of further named frames of the model (RobotModel.named_frames, e.g. the hand frames) to write the same way."""
    f = lambda v: " ".join(repr(float(t)) for t in v)  # noqa: E731
    lname = [root_link] + [model.joint_names[j] + "_link" for j in range(model.NDoF)]
    out = [f'<robot name="{name}">']
    for l in range(model.NDoF + 1):
        I = model.inertia[l]
        out.append(f'  <link name="{lname[l]}"><inertial><origin xyz="{f(model.com[l])}" rpy="0 0 0"/><mass value="{float(model.mass[l])!r}"/>'
                   f'<inertia ixx="{float(I[0, 0])!r}" ixy="{float(I[0, 1])!r}" ixz="{float(I[0, 2])!r}" iyy="{float(I[1, 1])!r}" iyz="{float(I[1, 2])!r}" izz="{float(I[2, 2])!r}"/></inertial></link>')
    for j in range(model.NDoF):
        lim = ""
        if model.min_joint_positions is not None and np.isfinite(model.min_joint_positions[j]):
            lim = f'<limit lower="{float(model.min_joint_positions[j])!r}" upper="{float(model.max_joint_positions[j])!r}" effort="100" velocity="10"/>'
        out.append(f'  <joint name="{model.joint_names[j]}" type="revolute"><parent link="{lname[int(model.parent[j])]}"/><child link="{lname[j + 1]}"/>'
                   f'<origin xyz="{f(model.o_fix[j])}" rpy="{f(rpy_from_rot(model.R_fix[j]))}"/><axis xyz="{f(model.axis[j])}"/>{lim}</joint>')
    for k, fname in enumerate(frames):
        out.append(f'  <link name="{fname}"/>')
        out.append(f'  <joint name="{fname}_fixed_joint" type="fixed"><parent link="{lname[int(model.frame_link[k])]}"/><child link="{fname}"/>'
                   f'<origin xyz="{f(model.frame_o[k])}" rpy="{f(rpy_from_rot(model.frame_R[k]))}"/></joint>')
    for fname in extra_frames:
        link, R, o = model.resolve_frame(fname)
        out.append(f'  <link name="{fname}"/>')
        out.append(f'  <joint name="{fname}_fixed_joint" type="fixed"><parent link="{lname[link]}"/><child link="{fname}"/>'
                   f'<origin xyz="{f(o)}" rpy="{f(rpy_from_rot(R))}"/></joint>')
    out.append("</robot>")
    return "\n".join(out)
