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


def poses_from_matrices(Ts) -> np.ndarray:
    """pose_from_matrix for a batch: (n, 4, 4) -> (n, 7), the same branches (Eigen's matrix -> quaternion) evaluated with masks —
    element for element what the scalar routine returns (the batched solvers' mirrors convert thousands of poses per call)."""
    m = np.asarray(Ts, dtype=np.float64).reshape(-1, 4, 4)
    n = m.shape[0]
    q = np.zeros((n, 4))
    d = np.stack([m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]], axis=1)
    tr = d[:, 0] + d[:, 1] + d[:, 2]
    pos = tr > 0
    if pos.any():
        t = np.sqrt(tr[pos] + 1.0)
        q[pos, 0] = 0.5 * t
        t = 0.5 / t
        mp = m[pos]
        q[pos, 1] = (mp[:, 2, 1] - mp[:, 1, 2]) * t
        q[pos, 2] = (mp[:, 0, 2] - mp[:, 2, 0]) * t
        q[pos, 3] = (mp[:, 1, 0] - mp[:, 0, 1]) * t
    neg = ~pos
    if neg.any():
        i = np.zeros(n, dtype=np.int64)
        i[d[:, 1] > d[:, 0]] = 1
        i[d[:, 2] > d[np.arange(n), i]] = 2
        for ii in range(3):
            sel = neg & (i == ii)
            if not sel.any():
                continue
            j, k = (ii + 1) % 3, (ii + 2) % 3
            ms = m[sel]
            t = np.sqrt(ms[:, ii, ii] - ms[:, j, j] - ms[:, k, k] + 1.0)
            q[sel, 1 + ii] = 0.5 * t
            t = 0.5 / t
            q[sel, 0] = (ms[:, k, j] - ms[:, j, k]) * t
            q[sel, 1 + j] = (ms[:, j, ii] + ms[:, ii, j]) * t
            q[sel, 1 + k] = (ms[:, k, ii] + ms[:, ii, k]) * t
    return np.concatenate([q, m[:, :3, 3]], axis=1)


def poses_to_matrices(P) -> np.ndarray:
    """pose_to_matrix for a batch: (n, 7) -> (n, 4, 4) (quaternions normalised, as restore_pose does)."""
    P = np.asarray(P, dtype=np.float64).reshape(-1, 7)
    q = P[:, :4] / np.linalg.norm(P[:, :4], axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    out = np.zeros((P.shape[0], 4, 4))
    out[:, 0, 0] = 1 - (tyy + tzz); out[:, 0, 1] = txy - twz; out[:, 0, 2] = txz + twy
    out[:, 1, 0] = txy + twz; out[:, 1, 1] = 1 - (txx + tzz); out[:, 1, 2] = tyz - twx
    out[:, 2, 0] = txz - twy; out[:, 2, 1] = tyz + twx; out[:, 2, 2] = 1 - (txx + tyy)
    out[:, :3, 3] = P[:, 4:]
    out[:, 3, 3] = 1.0
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
