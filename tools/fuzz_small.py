#!/usr/bin/env python3
"""Randomised parity sweep of the per-view wave solvers (homography, planar pose) and the semi-DLT solve against the CPU oracle.
usage: python tools/fuzz_small.py [n_cases] [seed]"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calibration_amd import capi, optim
from tests import synth
from calibration_amd.capi import CbaSummary, dptr
from calibration_amd.geometry import pose_to_matrix
from tests import helpers

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
orc = helpers.load_oracle()
lib = capi.load_library()
lib.cba_optimize_intrinsics_semidlt.argtypes = helpers.SEMIDLT_SOLVE_ARGS
out, t0 = {}, time.time()

# ---- homography: one batch of n_cases ragged views --------------------------------------------------------------------------
views, inits, deltas = [], [], []
for i in range(n_cases):
    n = int(rng.integers(4, 600))
    view, _ = helpers.homography_scene(n, float(rng.choice([0.0, 0.2, 1.0])), n_outliers=int(rng.integers(0, max(1, n // 8))), seed=int(rng.integers(1, 1 << 20)))
    H0 = helpers.dlt_homography(view[:n]) * (1 + 1e-3)
    H0[2, 2] = 1.0
    views.append(view); inits.append(H0)
bad = 0
for delta in (1.0, -1.0):
    res = optim.optimize_homography_batch(views, inits, optim.OptimOptions(huber_delta=delta))
    o = helpers.options(huber_delta=delta)
    for view, H0, r in zip(views, inits, res):
        X, Y, u, v = (np.ascontiguousarray(view[:, k]) for k in range(4))
        h, s = H0.reshape(9).copy(), CbaSummary()
        orc.orc_homography_solve(len(view), dptr(X), dptr(Y), dptr(u), dptr(v), dptr(h), C.byref(o), C.byref(s), dptr(None))
        ok = bool(s.success) == r.core.success and abs(s.iterations - r.core.iterations) <= 2 and \
            abs(s.final_cost - r.core.final_cost) <= 1e-8 * max(1.0, s.final_cost) and \
            np.abs(h.reshape(3, 3) - r.homography).max() <= 1e-7 * max(1.0, np.abs(h).max())
        bad += not ok
out["homography"] = dict(cases=2 * n_cases, disagreements=int(bad))

def _pose6_matrix(p6):
    """[R | t] (3 x 4) of an angle-axis + translation 6-vector (Rodrigues)"""
    w = np.asarray(p6[:3], float)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) if th == 0 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / (th * th) * (K @ K)
    return np.c_[R, np.asarray(p6[3:6], float)]


# ---- planar pose: one batch ---------------------------------------------------------------------------------------------------
cam = synth.camera_gt(0, distortion=True)
views, inits, nrs = [], [], []
for i in range(n_cases):
    rows, cols = int(rng.integers(3, 12)), int(rng.integers(3, 12))
    T = synth.random_view_poses(1, rng, dist=float(rng.uniform(0.5, 2.0)), max_tilt_deg=35.0, jitter=0.1)[0]
    views.append(synth.render_view(cam, T, synth.make_target_grid(rows, cols, 0.05), float(rng.choice([0.0, 0.2])), rng, cull=False))
    inits.append(synth.perturb_pose(T, rng, rot_deg=2.0, trans=0.01))
bad, bad_pp = 0, []
for nr in (0, 1, 2, 3):
    res = optim.optimize_planar_pose_batch(views, cam[:5], inits, optim.PlanarPoseOptions(num_radial=nr))
    o = helpers.options()
    for view, T0, r in zip(views, inits, res):
        X, Y, u, v = (np.ascontiguousarray(view[:, k]) for k in range(4))
        p, s, d, rms = helpers.pose6_of(T0), CbaSummary(), np.zeros(nr + 2), C.c_double()
        st = orc.orc_planar_pose_solve(len(view), dptr(X), dptr(Y), dptr(u), dptr(v), dptr(np.ascontiguousarray(cam[:5])), nr, dptr(p), C.byref(o),
                                       C.byref(s), dptr(d), C.byref(rms), dptr(None))
        if st != 0:  # < 8 points: the oracle's block refuses like fit_distortion_full; the product reports FAILURE
            bad += r.core.success
            continue
        # compared as TRANSFORMS: an angle-axis vector is not unique (theta and theta - 2 pi about the opposite axis are one rotation),
        # and the oracle returns whatever its iteration produced while r.pose went through a rotation matrix
        pose_gap = float(np.abs(_pose6_matrix(p) - np.asarray(r.pose)[:3, :4]).max())
        ok = bool(s.success) == r.core.success and abs(s.iterations - r.core.iterations) <= 2 and pose_gap <= 1e-6 and \
            abs(rms.value - r.reprojection_error) <= 1e-8
        bad += not ok
        if not ok:
            bad_pp.append(dict(nr=nr, n_points=len(view), success=[bool(s.success), r.core.success], iters=[int(s.iterations), int(r.core.iterations)],
                               pose_diff=pose_gap, rms=[rms.value, r.reprojection_error]))
out["planar_pose"] = dict(cases=4 * n_cases, disagreements=int(bad), bad=bad_pp[:8])

# ---- semi-DLT: 4..6 views, random options ----------------------------------------------------------------------------------------
bad, worst = 0, 0.0
for i in range(n_cases):
    nr = int(rng.integers(0, 4))
    d, kgt, agt = helpers.semidlt_scene(int(rng.integers(4, 7)), rows=int(rng.integers(4, 9)), cols=int(rng.integers(5, 10)),
                                        noise=float(rng.choice([0.0, 0.2])), nr=nr, seed=int(rng.integers(1, 1 << 20)))
    o = helpers.options(epsilon=1e-12, optimize_skew=int(rng.integers(0, 2)), huber_delta=float(rng.choice([1.0, -1.0])))
    a = helpers.semidlt_solve(orc.orc_semidlt_solve, d, nr, o, want_cov=False)
    b = helpers.semidlt_solve(lib.cba_optimize_intrinsics_semidlt, d, nr, o, want_cov=False)
    pd = max(np.abs(a[1] - b[1]).max() / np.abs(a[1]).max(), np.abs(a[2] - b[2]).max())
    ok = a[0] == b[0] == 0 and a[3].termination == b[3].termination and abs(a[3].iterations - b[3].iterations) <= 2 and \
        abs(a[3].final_cost - b[3].final_cost) <= 1e-8 * max(1.0, a[3].final_cost) + 1e-14 and pd <= 1e-5
    bad += not ok
    worst = max(worst, pd)
out["semidlt"] = dict(cases=n_cases, disagreements=int(bad), worst_param_diff=float(worst))
out["seconds"] = time.time() - t0
print(json.dumps(out))
