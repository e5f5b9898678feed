"""Stand-in for adam.casadi.KinDynComputations on the SYNTHETIC robot model (hippopt_amd/robot_model.py), using the
conventions of SURVEY Appendix A.  adam-robotics and the ergoCub URDF are not installable here; this exists only so the
reference's planner code can be executed by tools/gen_planner_fixtures.py."""
import casadi as cs
import numpy as np

STANDIN_MODEL = None  # set by the fixture generator (a hippopt_amd.robot_model.RobotModel)


def _rot_axis(axis, q):
    a = np.asarray(axis, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    aa = np.outer(a, a)
    return cs.cos(q) * cs.DM(np.eye(3) - aa) + cs.sin(q) * cs.DM(K) + cs.DM(aa)


class _Algos:
    def __init__(self, model):
        self.model = model


class KinDynComputations:
    def __init__(self, urdfstring=None, joints_name_list=None, root_link="root_link", gravity=None, f_opts=None):
        self.md = STANDIN_MODEL
        assert self.md is not None, "set adam.casadi.STANDIN_MODEL first"
        self.NDoF = self.md.NDoF
        self.g = np.array(gravity, float) if gravity is not None else np.array([0, 0, -9.80665, 0, 0, 0.0])
        self.rbdalgos = _Algos(self.md)
        self.f_opts = f_opts

    def get_total_mass(self):
        return self.md.get_total_mass()

    def _poses(self, H, s, links=None):
        md = self.md
        R = {0: H[:3, :3]}
        o = {0: H[:3, 3]}
        for j in range(md.NDoF):
            if links is not None and (j + 1) not in links:
                continue
            par = int(md.parent[j])
            R[j + 1] = cs.mtimes(R[par], cs.mtimes(cs.DM(md.R_fix[j]), _rot_axis(md.axis[j], s[j])))
            o[j + 1] = o[par] + cs.mtimes(R[par], cs.DM(md.o_fix[j]))
        return R, o

    def _chain(self, link):
        out = {0}
        while link > 0:
            out.add(link)
            link = int(self.md.parent[link - 1])
        return out

    def forward_kinematics_fun(self, frame):
        from hippopt_amd.robot_model import FRAME_NAMES
        md = self.md
        if frame in FRAME_NAMES:
            f = FRAME_NAMES.index(frame)
            link, fR, fo = int(md.frame_link[f]), md.frame_R[f], md.frame_o[f]
        else:   # any other named frame of the model (e.g. the hand frames of the pose finder)
            link, fR, fo = md.resolve_frame(frame)
        H = cs.MX.sym("H", 4, 4)
        s = cs.MX.sym("s", md.NDoF)
        R, o = self._poses(H, s, self._chain(link))
        Rf = cs.mtimes(R[link], cs.DM(fR))
        of = o[link] + cs.mtimes(R[link], cs.DM(fo))
        T = cs.vertcat(cs.horzcat(Rf, of), cs.DM([[0.0, 0.0, 0.0, 1.0]]))
        return cs.Function("T_fk", [H, s], [T])

    def CoM_position_fun(self):  # noqa: N802
        md = self.md
        H = cs.MX.sym("H", 4, 4)
        s = cs.MX.sym("s", md.NDoF)
        R, o = self._poses(H, s)
        acc = cs.DM.zeros(3, 1)
        for l in range(md.NDoF + 1):
            acc = acc + float(md.mass[l]) * (o[l] + cs.mtimes(R[l], cs.DM(md.com[l])))
        return cs.Function("CoM_pos", [H, s], [acc / md.get_total_mass()])

    def centroidal_momentum_matrix_fun(self):
        """6 x (6+NDoF) matrix A_G with h_G = A_G [v_b; omega; s_dot] (mixed representation): column k is the
        centroidal momentum produced by the unit velocity e_k."""
        md = self.md
        H = cs.MX.sym("H", 4, 4)
        s = cs.MX.sym("s", md.NDoF)
        R, o = self._poses(H, s)
        M = md.get_total_mass()
        c = {l: o[l] + cs.mtimes(R[l], cs.DM(md.com[l])) for l in range(md.NDoF + 1)}
        com = cs.DM.zeros(3, 1)
        for l in c:
            com = com + (float(md.mass[l]) / M) * c[l]
        Iw = {l: cs.mtimes(cs.mtimes(R[l], cs.DM(md.inertia[l])), R[l].T) for l in c}
        a = {j: cs.mtimes(R[j + 1], cs.DM(md.axis[j])) for j in range(md.NDoF)}
        # subtree membership
        anc = {}
        for l in range(md.NDoF + 1):
            p, q = set(), l
            while q > 0:
                p.add(q - 1)
                q = int(md.parent[q - 1])
            anc[l] = p
        cols = []
        for k in range(6 + md.NDoF):
            lin = cs.DM.zeros(3, 1)
            ang = cs.DM.zeros(3, 1)
            for l in c:
                if k < 3:
                    w, v = None, cs.DM(np.eye(3)[:, k])
                elif k < 6:
                    e = cs.DM(np.eye(3)[:, k - 3])
                    w, v = e, cs.cross(e, c[l] - o[0])
                else:
                    j = k - 6
                    if j not in anc[l]:
                        continue
                    w, v = a[j], cs.cross(a[j], c[l] - o[j + 1])
                lin = lin + float(md.mass[l]) * v
                ang = ang + float(md.mass[l]) * cs.cross(c[l] - com, v)
                if w is not None:
                    ang = ang + cs.mtimes(Iw[l], w)
            cols.append(cs.vertcat(lin, ang))
        return cs.Function("CMM", [H, s], [cs.horzcat(*cols)])
