"""numpy restatement of the reference's per-view pose seed, used only to build initial guesses for
the KAT tests the way the reference's tests do (intrinsics_optimize_test.cpp:40-46 calls
estimate_planar_pose(view, guess_cam.kmtx)): Hartley-normalised DLT homography
(src/estimation/linear/homographyestimator.cpp:123-174) followed by pose-from-homography
(src/estimation/linear/planarpose_linear.cpp:17-76).  Host seed code: out of the hot-path scope."""
import numpy as np


def _normalise(p):
    c = p.mean(axis=0)
    d = np.sqrt(((p - c) ** 2).sum(axis=1)).mean()
    s = np.sqrt(2.0) / d
    T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    return (p - c) * s, T


def homography_dlt(src, dst):
    a, Ta = _normalise(src)
    b, Tb = _normalise(dst)
    rows = []
    for (x, y), (u, v) in zip(a, b):
        rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    _, _, vt = np.linalg.svd(np.asarray(rows))
    H = vt[-1].reshape(3, 3)
    H = np.linalg.inv(Tb) @ H @ Ta
    return H / H[2, 2]


def estimate_planar_pose(view, kmtx5):
    """view: (N,4) [X,Y,u,v]; kmtx5 = [fx,fy,cx,cy,skew] -> 4x4 c_T_t."""
    fx, fy, cx, cy, skew = kmtx5
    v = np.asarray(view)
    yn = (v[:, 3] - cy) / fy
    xn = (v[:, 2] - cx - skew * yn) / fx
    H = homography_dlt(v[:, :2], np.stack([xn, yn], axis=1))
    h1, h2, h3 = H[:, 0], H[:, 1], H[:, 2]
    s = 2.0 / (np.linalg.norm(h1) + np.linalg.norm(h2))
    r1, r2, t = s * h1, s * h2, s * h3
    if t[2] < 0:
        r1, r2, t = -r1, -r2, -t
    R = np.stack([r1, r2, np.cross(r1, r2)], axis=1)
    U, _, Vt = np.linalg.svd(R)
    R = U @ np.diag([1, 1, np.linalg.det(U @ Vt)]) @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T
