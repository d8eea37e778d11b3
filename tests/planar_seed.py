"""numpy restatement of the reference's per-view pose seed (TEST INFRASTRUCTURE: the checker for the batched GPU seed
cba_estimate_planar_pose_batch, and the initial guesses of the KAT tests, which the reference's tests build the same way —
intrinsics_optimize_test.cpp:40-46 calls estimate_planar_pose(view, guess_cam.kmtx)):
  estimate_planar_pose                     src/estimation/linear/planarpose_linear.cpp:54-76
  normalize (pixel -> normalised)          include/calib/models/camera_matrix.h:33-39
  Hartley-normalised DLT homography        src/estimation/linear/homographyestimator.cpp:17-87 (Eigen::JacobiSVD -> numpy SVD)
  pose_from_homography_normalized          src/estimation/linear/planarpose_linear.cpp:17-52
"""
import numpy as np


def _normalise(p):
    """normalize_points_2d (homographyestimator.cpp:17-45)"""
    c = p.mean(axis=0)
    d = np.sqrt(((p - c) ** 2).sum(axis=1)).mean()
    s = np.sqrt(2.0) / d if d > 0 else 1.0
    T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    return np.c_[s * p[:, 0] - s * c[0], s * p[:, 1] - s * c[1]], T


def homography_dlt(src, dst):
    a, Ta = _normalise(np.asarray(src, dtype=float))
    b, Tb = _normalise(np.asarray(dst, dtype=float))
    rows = []
    for (x, y), (u, v) in zip(a, b):
        rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    _, _, vt = np.linalg.svd(np.asarray(rows))
    H = vt[-1].reshape(3, 3)
    H = H / H[2, 2]
    return np.linalg.inv(Tb) @ H @ Ta


def pose_from_homography_normalized(H):
    h1, h2, h3 = H[:, 0], H[:, 1], H[:, 2]
    s = np.sqrt(np.linalg.norm(h1) * np.linalg.norm(h2))
    if s < 1e-12:
        s = 1.0
    r1, r2 = h1 / s, h2 / s
    Ri = np.stack([r1, r2, np.cross(r1, r2)], axis=1)
    U, _, Vt = np.linalg.svd(Ri)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        V = Vt.T.copy()
        V[:, 2] *= -1.0
        R = U @ V.T
    t = h3 / s
    if R[2, 2] < 0:
        R, t = -R, -t
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def estimate_planar_pose(view, kmtx5):
    """view: (N,4) [X,Y,u,v]; kmtx5 = [fx,fy,cx,cy,skew] -> 4x4 c_T_t."""
    fx, fy, cx, cy, skew = np.asarray(kmtx5, dtype=float).reshape(-1)[:5]
    v = np.asarray(view, dtype=float).reshape(-1, 4)
    if len(v) < 4:
        return np.eye(4)
    yn = (v[:, 3] - cy) / fy
    xn = (v[:, 2] - cx - skew * yn) / fx
    H = homography_dlt(v[:, :2], np.stack([xn, yn], axis=1))
    if abs(H[2, 2]) > 1e-15:
        H = H / H[2, 2]
    return pose_from_homography_normalized(H)
