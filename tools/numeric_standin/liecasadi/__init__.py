"""NUMERIC stand-in for the liecasadi classes the reference's host-side utilities use (SE3 / SO3 / Quaternion on numbers,
xyzw order).  Restated from liecasadi's published formulas (slerp step: the Wikipedia quaternion slerp liecasadi cites).
Used ONLY by tools/gen_interpolator_fixtures.py; never shipped."""
import numpy as np


def _col(x):
    return np.asarray(x, dtype=float).reshape(-1, 1)


class Quaternion:
    def __init__(self, xyzw):
        self.xyzw = _col(xyzw)

    def coeffs(self):
        return self.xyzw

    @staticmethod
    def slerp_step(q1, q2, t):
        q1, q2 = _col(q1), _col(q2)
        dot = float(np.sum(q1 * q2))
        angle = np.arccos(dot)
        with np.errstate(divide="ignore", invalid="ignore"):
            return Quaternion((np.sin((1.0 - t) * angle) * q1 + np.sin(t * angle) * q2) / np.sin(angle))


class SO3:
    def __init__(self, xyzw):
        self.xyzw = _col(xyzw)

    @staticmethod
    def Identity():  # noqa: N802
        return SO3([0.0, 0.0, 0.0, 1.0])

    @staticmethod
    def from_quat(xyzw):
        return SO3(xyzw)

    def as_quat(self):
        return Quaternion(self.xyzw)

    def as_matrix(self):
        x, y, z, w = self.xyzw.reshape(-1)
        K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
        return np.eye(3) + 2 * w * K + 2 * K @ K

    def act(self, p):
        return self.as_matrix() @ _col(p)


class SE3:
    def __init__(self, pos, xyzw):
        self.pos, self.xyzw = _col(pos), _col(xyzw)

    @staticmethod
    def from_position_quaternion(pos, xyzw):
        return SE3(pos, xyzw)

    @staticmethod
    def from_translation_and_rotation(translation, rotation):
        return SE3(translation, rotation.xyzw)

    def translation(self):
        return self.pos

    def rotation(self):
        return SO3(self.xyzw)
