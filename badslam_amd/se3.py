"""Minimal SE(3) helpers (numpy, float64) used by the synthetic-scene generator, tests and bench.

Poses are stored the way the reference stores ``SE3f`` (Sophus): unit quaternion ``(x, y, z, w)``
plus translation, packed as a length-7 array ``[qx, qy, qz, qw, tx, ty, tz]``.  Tangent vectors are
``[upsilon(3), omega(3)]`` (libvis/third_party/sophus/sophus/se3.hpp:293-313).
"""
import numpy as np


def identity():
    return np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64)


def _hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def quat_to_rot(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def rot_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    q = np.array(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def exp(xi):
    xi = np.asarray(xi, dtype=np.float64)
    u, w = xi[:3], xi[3:]
    theta = np.linalg.norm(w)
    W = _hat(w)
    if theta < 1e-10:
        R = np.eye(3) + W
        V = np.eye(3) + 0.5 * W
    else:
        R = np.eye(3) + np.sin(theta) / theta * W + (1 - np.cos(theta)) / theta ** 2 * (W @ W)
        V = np.eye(3) + (1 - np.cos(theta)) / theta ** 2 * W + (theta - np.sin(theta)) / theta ** 3 * (W @ W)
    return np.concatenate([rot_to_quat(R), V @ u])


def log(T):
    R = quat_to_rot(T[:4])
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    theta = np.arccos(c)
    if theta < 1e-10:
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
        Vinv = np.eye(3) - 0.5 * _hat(w)
    else:
        w = theta / (2 * np.sin(theta)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        W = _hat(w)
        Vinv = np.eye(3) - 0.5 * W + (1 - theta * np.cos(theta / 2) / (2 * np.sin(theta / 2))) / theta ** 2 * (W @ W)
    return np.concatenate([Vinv @ T[4:], w])


def matrix(T):
    M = np.eye(4)
    M[:3, :3] = quat_to_rot(T[:4])
    M[:3, 3] = T[4:]
    return M


def from_matrix(M):
    return np.concatenate([rot_to_quat(M[:3, :3]), M[:3, 3]])


def mul(a, b):
    return from_matrix(matrix(a) @ matrix(b))


def inverse(T):
    R = quat_to_rot(T[:4])
    return np.concatenate([rot_to_quat(R.T), -R.T @ T[4:]])
