#!/usr/bin/env python3
"""Generates tests/golden/vp_jacobians.json (run from the repo root: ``python tests/golden/gen_golden_vp.py``): residuals and
Jacobians of the three small-solver functors by the COMPLEX-STEP method on an independent numpy restatement — the derivative
THROUGH the linear least-squares solve of the variable-projection functors is obtained by running the normal equations in
complex arithmetic (an analytic function of the parameters), not by any hand-derived Golub-Pereyra formula:

  planar_pose   PlanarPoseVPResidual (src/estimation/optim/planarpose.cpp:39-57; to_observation observationutils.h:97-113 with
                ceres::AngleAxisRotatePoint; fit_distortion_full include/calib/models/distortion.h:229-295)
  semidlt       CalibVPResidual (src/estimation/residuals/intrinsicsemidltresidual.h:34-58; planar_observables_to_observables
                observationutils.h:78-95 with Eigen's un-normalised quaternion -> rotation)
  homography    HomographyResidual (src/estimation/optim/homography.cpp:103-130)

Shares no code with oracle/ (dual numbers) nor with the product's analytic derivatives."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
H = 1e-30


def design(K, x, y, u, v, nr):
    fx, fy, cx, cy, skew = K
    r2 = x * x + y * y
    cols_u, cols_v, rp = [], [], r2
    for _ in range(nr):
        cols_u.append(fx * x * rp + skew * y * rp)
        cols_v.append(fy * y * rp)
        rp = rp * r2
    cols_u += [fx * (2 * x * y) + skew * (r2 + 2 * y * y), fx * (r2 + 2 * x * x) + skew * (2 * x * y)]
    cols_v += [fy * (r2 + 2 * y * y), fy * (2 * x * y)]
    A = np.empty((2 * len(x), nr + 2), dtype=complex)
    A[0::2] = np.stack(cols_u, axis=1)
    A[1::2] = np.stack(cols_v, axis=1)
    b = np.empty(2 * len(x), dtype=complex)
    b[0::2] = u - (fx * x + skew * y + cx)
    b[1::2] = v - (fy * y + cy)
    return A, b


def vp_residual(A, b):
    alpha = np.linalg.solve(A.T @ A, A.T @ b)  # plain transpose: analytic in the parameters
    return A @ alpha - b, alpha


def aa_rotate(aa, P):
    th2 = aa @ aa
    th = np.sqrt(th2)
    w = aa / th
    c, s = np.cos(th), np.sin(th)
    return P * c + np.cross(w, P) * s + np.outer(P @ w, w) * (1 - c)


def planar_residual(pose6, K, X, Y, u, v, nr):
    P = np.stack([X, Y, np.zeros_like(X)], axis=1).astype(complex)
    Pc = aa_rotate(pose6[:3], P) + pose6[3:]
    return vp_residual(*design(K, Pc[:, 0] / Pc[:, 2], Pc[:, 1] / Pc[:, 2], u, v, nr))


def quat_R(q):  # Eigen::Quaternion::toRotationMatrix, no normalisation
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def semidlt_residual(theta, views, nr):
    K = theta[:5]
    xs, ys, us, vs = [], [], [], []
    for i, (X, Y, u, v) in enumerate(views):
        q, t = theta[5 + 7 * i:9 + 7 * i], theta[9 + 7 * i:12 + 7 * i]
        R = quat_R(q)
        Pc = np.stack([X, Y, np.zeros_like(X)], axis=1).astype(complex) @ R.T + t
        xs.append(Pc[:, 0] / Pc[:, 2]); ys.append(Pc[:, 1] / Pc[:, 2]); us.append(u); vs.append(v)
    return vp_residual(*design(K, np.concatenate(xs), np.concatenate(ys), np.concatenate(us), np.concatenate(vs), nr))


def cstep(f, theta):
    theta = np.asarray(theta, dtype=complex)
    r0, alpha = f(theta)
    J = np.empty((len(r0), len(theta)))
    for k in range(len(theta)):
        t = theta.copy()
        t[k] += 1j * H
        J[:, k] = f(t)[0].imag / H
    return r0.real, J, alpha.real


def main():
    from tests import synth
    from calibration_amd.geometry import pose_from_matrix

    rng = np.random.default_rng(20261004)
    out = {"planar_pose": [], "semidlt": [], "homography": []}
    cam = synth.camera_gt(0, distortion=True)
    for nr in (0, 1, 2, 3):
        grid = synth.make_target_grid(5, 6, 0.07)
        T = synth.random_view_poses(1, rng, dist=1.0, max_tilt_deg=30.0, jitter=0.1)[0]
        view = synth.render_view(cam, T, grid, 0.3, rng, cull=False)
        K = cam[:5] * (1 + 0.01 * rng.uniform(-1, 1, 5))
        K[4] = 0.3
        q = pose_from_matrix(synth.perturb_pose(T, rng))
        ang = 2 * np.arctan2(np.linalg.norm(q[1:4]), q[0])
        pose6 = np.concatenate([q[1:4] / np.linalg.norm(q[1:4]) * ang, q[4:]])
        X, Y, u, v = (view[:, k].copy() for k in range(4))
        r, J, al = cstep(lambda th: planar_residual(th, K, X, Y, u, v, nr), pose6)
        out["planar_pose"].append(dict(nr=nr, K=K.tolist(), X=X.tolist(), Y=Y.tolist(), u=u.tolist(), v=v.tolist(), pose6=pose6.tolist(),
                                       r=r.tolist(), J=J.tolist(), alpha=al.tolist()))
    for nr, nv in ((1, 3), (2, 4), (3, 3)):
        grid = synth.make_target_grid(4, 5, 0.09)
        poses = synth.random_view_poses(nv, rng, dist=1.0, max_tilt_deg=30.0, jitter=0.1)
        views, th = [], [cam[:5] * (1 + 0.01 * rng.uniform(-1, 1, 5))]
        th[0][4] = -0.2
        for T in poses:
            vw = synth.render_view(cam, T, grid, 0.3, rng, cull=False)
            views.append(tuple(vw[:, k].copy() for k in range(4)))
            p7 = pose_from_matrix(synth.perturb_pose(T, rng))
            p7[:4] *= 1.0 + 0.05 * rng.uniform(-1, 1)  # un-normalised quaternion: the functor does not normalise
            th.append(p7)
        theta = np.concatenate(th)
        r, J, al = cstep(lambda t: semidlt_residual(t, views, nr), theta)
        out["semidlt"].append(dict(nr=nr, n_views=nv, views=[[a.tolist() for a in vw] for vw in views], kappa=theta[:5].tolist(),
                                   poses7=theta[5:].reshape(nv, 7).tolist(), r=r.tolist(), J=J.tolist(), alpha=al.tolist()))
    h = np.array([0.995, -0.0998, 10.0, 0.0998, 0.995, -5.0, 0.001, -0.002]) * (1 + 0.01 * rng.uniform(-1, 1, 8))
    pts = rng.uniform(-100, 100, (6, 4))
    rows = []
    for x, y, u, v in pts:
        def f(hc, x=x, y=y, u=u, v=v):
            w = hc[6] * x + hc[7] * y + 1
            return np.array([(hc[0] * x + hc[1] * y + hc[2]) / w - u, (hc[3] * x + hc[4] * y + hc[5]) / w - v]), np.zeros(1)
        r, J, _ = cstep(f, h)
        rows.append(dict(x=x, y=y, u=u, v=v, r=r.tolist(), J=J.tolist()))
    out["homography"] = dict(h=h.tolist(), points=rows)
    path = os.path.join(HERE, "vp_jacobians.json")
    json.dump(out, open(path, "w"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
