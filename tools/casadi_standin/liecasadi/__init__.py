"""Stand-in for the few liecasadi classes hippopt touches (xyzw quaternions), on top of the CasADi stand-in.
Restated from liecasadi's published formulas; see tools/casadi_standin/casadi/__init__.py for the purpose."""
import casadi as cs


class Quaternion:
    def __init__(self, xyzw):
        self.xyzw = cs.MX._wrap(xyzw)

    def coeffs(self):
        return self.xyzw

    def normalize(self):
        return Quaternion(xyzw=self.xyzw / cs.norm_2(self.xyzw))

    @staticmethod
    def product(a, b):
        av, aw, bv, bw = a[:3], a[3], b[:3], b[3]
        return cs.vertcat(aw * bv + bw * av + cs.cross(av, bv), aw * bw - cs.mtimes(av.T, bv))

    def __mul__(self, other):
        return Quaternion(xyzw=Quaternion.product(self.xyzw, other.xyzw))

    def __sub__(self, other):
        return Quaternion(xyzw=self.xyzw - other.xyzw)

    def __add__(self, other):
        return Quaternion(xyzw=self.xyzw + other.xyzw)


class SO3:
    def __init__(self, xyzw):
        self.xyzw = cs.MX._wrap(xyzw)
        self.quat = Quaternion(self.xyzw)

    @staticmethod
    def from_quat(xyzw):
        return SO3(xyzw)

    @staticmethod
    def Identity():  # noqa: N802
        return SO3(cs.DM([0.0, 0.0, 0.0, 1.0]))

    def as_quat(self):
        return self.quat

    def as_matrix(self):
        v, w = self.xyzw[:3], self.xyzw[3]
        K = cs.skew(v)
        return cs.DM.eye(3) + 2 * w * K + 2 * cs.mtimes(K, K)

    def inverse(self):
        return SO3(cs.vertcat(-self.xyzw[:3], self.xyzw[3]))

    def __mul__(self, other):
        return SO3((self.quat * other.quat).coeffs())

    def act(self, p):
        return cs.mtimes(self.as_matrix(), p)


class SE3:
    def __init__(self, pos, xyzw):
        self.pos, self.xyzw = cs.MX._wrap(pos), cs.MX._wrap(xyzw)

    @staticmethod
    def from_position_quaternion(pos, xyzw):
        return SE3(pos, xyzw)

    def rotation(self):
        return SO3(self.xyzw)

    def translation(self):
        return self.pos

    def as_matrix(self):
        R = SO3(self.xyzw).as_matrix()
        return cs.vertcat(cs.horzcat(R, self.pos), cs.DM([[0.0, 0.0, 0.0, 1.0]]))
