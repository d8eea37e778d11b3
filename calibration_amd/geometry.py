"""Small host-side SE(3) helpers (pure numpy; no arithmetic of the hot path lives here).

They restate the conversions the reference performs on the host before/after calling Ceres:
``populate_quat_tran`` / ``restore_pose`` (src/estimation/detail/observationutils.h:43-62), i.e.
Eigen's matrix->quaternion and quaternion->matrix conversions, plus the test-side pose builders of
tests/unit/utils.h:54-64.
"""
from __future__ import annotations

import numpy as np


def quat_to_rotmat(q) -> np.ndarray:
    """Eigen::Quaternion(w,x,y,z).toRotationMatrix() — no normalisation (observationutils.h:20-24)."""
    w, x, y, z = (q[0], q[1], q[2], q[3])
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array(
        [[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]],
        dtype=np.result_type(w, x, y, z, np.float64),
    )


def rotmat_to_quat(m) -> np.ndarray:
    """Eigen::Quaterniond(Matrix3d) (Eigen/src/Geometry/Quaternion.h), storage (w,x,y,z)."""
    m = np.asarray(m, dtype=np.float64)
    q = np.zeros(4)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (m[2, 1] - m[1, 2]) * t
        q[2] = (m[0, 2] - m[2, 0]) * t
        q[3] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[1 + i] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[k, j] - m[j, k]) * t
        q[1 + j] = (m[j, i] + m[i, j]) * t
        q[1 + k] = (m[k, i] + m[i, k]) * t
    return q


def pose_from_matrix(m) -> np.ndarray:
    """populate_quat_tran (observationutils.h:43-48): 4x4 -> [qw,qx,qy,qz,tx,ty,tz]."""
    m = np.asarray(m, dtype=np.float64).reshape(4, 4)
    return np.concatenate([rotmat_to_quat(m[:3, :3]), m[:3, 3]])


def pose_to_matrix(p) -> np.ndarray:
    """restore_pose (observationutils.h:50-62): normalises the quaternion."""
    p = np.asarray(p, dtype=np.float64).reshape(7)
    q = p[:4] / np.linalg.norm(p[:4])
    out = np.eye(4)
    out[:3, :3] = quat_to_rotmat(q)
    out[:3, 3] = p[4:]
    return out


def quat_plus(q, d) -> np.ndarray:
    """ceres::QuaternionManifold::Plus: q_d (x) q, q_d = [cos|d|, sin|d|/|d| d]."""
    q = np.asarray(q, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64)
    n = np.linalg.norm(d)
    if n == 0:
        return q.copy()
    s = np.sin(n) / n
    a = np.array([np.cos(n), s * d[0], s * d[1], s * d[2]])
    return np.array(
        [
            a[0] * q[0] - a[1] * q[1] - a[2] * q[2] - a[3] * q[3],
            a[0] * q[1] + a[1] * q[0] + a[2] * q[3] - a[3] * q[2],
            a[0] * q[2] - a[1] * q[3] + a[2] * q[0] + a[3] * q[1],
            a[0] * q[3] + a[1] * q[2] - a[2] * q[1] + a[3] * q[0],
        ]
    )


def axis_angle_to_R(axis, angle) -> np.ndarray:
    """tests/unit/utils.h:54-57 (Eigen::AngleAxisd(angle, axis.normalized()).toRotationMatrix())."""
    if angle < 1e-16:
        return np.eye(3)
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def make_pose(t, axis, angle) -> np.ndarray:
    """tests/unit/utils.h:59-64."""
    T = np.eye(4)
    T[:3, :3] = axis_angle_to_R(axis, angle)
    T[:3, 3] = t
    return T


def inv(T) -> np.ndarray:
    T = np.asarray(T)
    out = np.eye(4)
    out[:3, :3] = T[:3, :3].T
    out[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return out


def rotation_angle(R) -> float:
    """tests/unit/utils.h:29-33."""
    c = (np.trace(R) - 1.0) * 0.5
    return float(np.arccos(max(-1.0, min(1.0, c))))
