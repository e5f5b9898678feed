"""Rigid transforms on numbers (xyzw quaternions): the part of liecasadi's SE3 / SO3 / Quaternion surface the reference's
host-side utilities use (robot_planning/utilities/interpolators.py, variables/contacts.py:103-127), on numpy."""
import numpy as np


def _col(x):
    return np.asarray(x, dtype=float).reshape(-1, 1)


class SO3:
    def __init__(self, xyzw):
        self.xyzw = _col(xyzw)

    @staticmethod
    def Identity():  # noqa: N802  (liecasadi name)
        return SO3([0.0, 0.0, 0.0, 1.0])

    @staticmethod
    def from_quat(xyzw):
        return SO3(xyzw)

    def as_quat(self):
        return self

    def coeffs(self):
        return self.xyzw

    def as_matrix(self):
        x, y, z, w = self.xyzw.reshape(-1)
        K = np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])
        return np.eye(3) + 2.0 * w * K + 2.0 * K @ K

    def act(self, p):
        return self.as_matrix() @ _col(p)


class SE3:
    def __init__(self, pos, xyzw):
        self.pos, self.xyzw = _col(pos), _col(xyzw)

    @staticmethod
    def from_position_quaternion(pos, xyzw):
        return SE3(pos, xyzw)

    @staticmethod
    def from_translation_and_rotation(translation, rotation: SO3):
        return SE3(translation, rotation.xyzw)

    def translation(self):
        return self.pos

    def rotation(self):
        return SO3(self.xyzw)


def slerp_step(q1, q2, t):
    """liecasadi.Quaternion.slerp_step: (sin((1-t) a) q1 + sin(t a) q2) / sin(a), a = acos(q1 . q2)."""
    q1, q2 = _col(q1), _col(q2)
    angle = np.arccos(float(np.sum(q1 * q2)))
    return (np.sin((1.0 - t) * angle) * q1 + np.sin(t * angle) * q2) / np.sin(angle)
