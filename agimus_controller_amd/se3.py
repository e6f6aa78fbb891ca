"""Minimal rigid-transform types for the host layer.

The reference passes `pinocchio.SE3` / `Motion` / `Force` objects in its trajectory points
(agimus_controller/agimus_controller/trajectory.py:5).  Pinocchio is optional here: any object
exposing `.rotation` / `.translation` (or a 7-vector xyz+quaternion) is accepted by the OCP
classes; these small numpy types cover the case where Pinocchio is not installed.
"""

from __future__ import annotations

import numpy as np


def quat_to_rot(q) -> np.ndarray:
    """Unit quaternion (x, y, z, w) -> rotation matrix."""
    x, y, z, w = np.asarray(q, dtype=float) / np.linalg.norm(q)
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ]
    )


def rot_to_quat(R) -> np.ndarray:
    """Rotation matrix -> unit quaternion (x, y, z, w), w >= 0 branch of Shepperd's method."""
    R = np.asarray(R, dtype=float)
    t = np.trace(R)
    if t > 0:
        s = 2.0 * np.sqrt(1.0 + t)
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = 2.0 * np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k])
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


class SE3:
    __slots__ = ("rotation", "translation")

    def __init__(self, rotation=None, translation=None):
        self.rotation = np.eye(3) if rotation is None else np.array(rotation, dtype=float).reshape(3, 3)
        self.translation = np.zeros(3) if translation is None else np.array(translation, dtype=float).reshape(3)

    @staticmethod
    def Identity() -> "SE3":
        return SE3()

    @staticmethod
    def Random(rng=None) -> "SE3":
        rng = np.random.default_rng() if rng is None else rng
        q = rng.normal(size=4)
        return SE3(quat_to_rot(q), rng.uniform(-1, 1, 3))

    def inverse(self) -> "SE3":
        return SE3(self.rotation.T, -self.rotation.T @ self.translation)

    def __mul__(self, other):
        if isinstance(other, SE3) or (hasattr(other, "rotation") and hasattr(other, "translation")):
            return SE3(self.rotation @ np.asarray(other.rotation), self.rotation @ np.asarray(other.translation) + self.translation)
        return self.rotation @ np.asarray(other, dtype=float) + self.translation

    def actInv(self, other):
        """self^-1 * other (pinocchio SE3.actInv on an SE3)."""
        return self.inverse() * other

    def act(self, other):
        return self * other

    def isIdentity(self, prec=1e-12) -> bool:
        return bool(np.allclose(self.rotation, np.eye(3), atol=prec) and np.allclose(self.translation, 0.0, atol=prec))

    def copy(self) -> "SE3":
        return SE3(self.rotation.copy(), self.translation.copy())

    @property
    def homogeneous(self) -> np.ndarray:
        H = np.eye(4)
        H[:3, :3], H[:3, 3] = self.rotation, self.translation
        return H

    def __eq__(self, other):
        return hasattr(other, "rotation") and np.array_equal(self.rotation, other.rotation) and np.array_equal(self.translation, other.translation)

    def __repr__(self):
        return f"SE3(R=\n{self.rotation},\n p={self.translation})"


class _Spatial6:
    __slots__ = ("vector",)

    def __init__(self, v=None):
        self.vector = np.zeros(6) if v is None else np.array(v, dtype=float).reshape(6)

    @property
    def linear(self):
        return self.vector[:3]

    @property
    def angular(self):
        return self.vector[3:]

    def __eq__(self, other):
        return hasattr(other, "vector") and np.array_equal(self.vector, other.vector)


class Motion(_Spatial6):
    pass


class Force(_Spatial6):
    pass


def log3(R) -> np.ndarray:
    """pinocchio.log3 (axis-angle vector of a rotation), with its branch near theta = pi."""
    R = np.asarray(R, dtype=float).reshape(3, 3)
    ct = 0.5 * (min(3.0, max(-1.0, float(np.trace(R)))) - 1.0)
    theta = np.arccos(ct)
    if theta >= np.pi - 1e-2:
        cphi = -ct
        beta = theta * theta / (1.0 + cphi)
        v = (np.diag(R) + cphi) * beta
        sgn = np.array([1.0 if R[2, 1] > R[1, 2] else -1.0, 1.0 if R[0, 2] > R[2, 0] else -1.0, 1.0 if R[1, 0] > R[0, 1] else -1.0])
        return sgn * np.sqrt(np.maximum(v, 0.0))
    t = 0.5 * (theta / np.sin(theta) if theta > 1e-8 else 1.0)
    return t * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def log6(M: "SE3") -> "Motion":
    """pinocchio.log6: Motion (linear, angular) with exp6(log6(M)) = M."""
    w = log3(M.rotation)
    p = np.asarray(M.translation, dtype=float)
    t2 = float(w @ w)
    t = np.sqrt(t2)
    if t2 < 1e-12:
        alpha, beta = 1.0 - t2 / 12.0, 1.0 / 12.0 + t2 / 720.0
    else:
        st, ct = np.sin(t), np.cos(t)
        i22 = 1.0 / (2.0 * (1.0 - ct))
        alpha, beta = t * st * i22, 1.0 / t2 - st / t * i22
    v = alpha * p - 0.5 * np.cross(w, p) + beta * float(w @ p) * w
    return Motion(np.concatenate([v, w]))


def XYZQUATToSE3(v) -> SE3:
    v = np.asarray(v, dtype=float).reshape(7)
    return SE3(quat_to_rot(v[3:]), v[:3])


def SE3ToXYZQUAT(M) -> np.ndarray:
    return np.concatenate([np.asarray(M.translation, dtype=float).reshape(3), rot_to_quat(M.rotation)])


def as_se3_12(pose) -> np.ndarray:
    """Anything pose-like -> 12 doubles (R row-major, then p): an object with .rotation/.translation
    (pinocchio.SE3 or the SE3 above), a 7-vector xyz+quat (what the reference's sine-wave generators
    emit, sine_wave_configuration_space.py:54), a 4x4 or a 12-vector."""
    if hasattr(pose, "rotation") and hasattr(pose, "translation"):
        return np.concatenate([np.asarray(pose.rotation, dtype=float).reshape(9), np.asarray(pose.translation, dtype=float).reshape(3)])
    a = np.asarray(pose, dtype=float)
    if a.size == 7:
        a = a.reshape(7)
        return np.concatenate([quat_to_rot(a[3:]).reshape(9), a[:3]])
    if a.shape == (4, 4):
        return np.concatenate([a[:3, :3].reshape(9), a[:3, 3]])
    if a.size == 12:
        return a.reshape(12).copy()
    raise ValueError(f"cannot interpret {pose!r} as a rigid transform")
