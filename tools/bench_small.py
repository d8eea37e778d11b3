#!/usr/bin/env python3
"""Throughput of the batched per-view solvers (one wavefront per view, whole LM in-kernel):
optimize_planar_pose_batch and optimize_homography_batch.  Prints one JSON line per case.
usage: python tools/bench_small.py [--views 8000] [--grid 8 11]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calibration_amd import optim  # noqa: E402
from tests import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=8000)
    ap.add_argument("--grid", type=int, nargs=2, default=[8, 11])
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    cam = synth.camera_gt(0, distortion=True)
    grid = synth.make_target_grid(args.grid[0], args.grid[1], 0.02)
    poses = synth.random_view_poses(args.views, rng, dist=0.6, max_tilt_deg=25.0, jitter=0.03)
    views = [synth.render_view(cam, T, grid, noise_px=0.2, rng=rng, cull=False) for T in poses]
    init = [synth.perturb_pose(T, rng, rot_deg=2.0, trans=0.01) for T in poses]
    n_pts = grid.shape[0]
    for rep in range(2):  # second pass: warm module + allocator
        t0 = time.perf_counter()
        res = optim.optimize_planar_pose_batch(views, cam[:5], init, optim.PlanarPoseOptions(num_radial=2))
        t_pp = time.perf_counter() - t0
    print(json.dumps({"case": "optimize_planar_pose_batch", "views": args.views, "points_per_view": n_pts, "wall_s": t_pp,
                      "kernel_plus_d2h_s": res[0].core.solve_seconds, "views_per_s": args.views / t_pp,
                      "converged": int(sum(r.core.success for r in res)),
                      "mean_iterations": float(np.mean([r.core.iterations for r in res])),
                      "median_rms_px": float(np.median([r.reprojection_error for r in res]))}))
    # homographies of the same views, seeded by a perturbed exact plane-to-image map
    K = np.array([[cam[0], cam[4], cam[2]], [0, cam[1], cam[3]], [0, 0, 1.0]])
    H0 = []
    for T in poses:
        H = K @ np.c_[T[:3, 0], T[:3, 1], T[:3, 3]]
        H = H / H[2, 2] * (1 + 1e-3 * rng.uniform(-1, 1, (3, 3)))
        H[2, 2] = 1.0
        H0.append(H)
    for rep in range(2):
        t0 = time.perf_counter()
        res = optim.optimize_homography_batch(views, H0)
        t_h = time.perf_counter() - t0
    print(json.dumps({"case": "optimize_homography_batch", "views": args.views, "points_per_view": n_pts, "wall_s": t_h,
                      "kernel_plus_d2h_s": res[0].core.solve_seconds, "views_per_s": args.views / t_h,
                      "converged": int(sum(r.core.success for r in res)),
                      "mean_iterations": float(np.mean([r.core.iterations for r in res]))}))


    # batched DLT seed of the same views
    for rep in range(2):
        t0 = time.perf_counter()
        seeds = optim.estimate_planar_pose_batch(views, cam[:5])
        t_s = time.perf_counter() - t0
    err = max(np.abs(S - T).max() for S, T in zip(seeds[:200], poses[:200]))
    print(json.dumps({"case": "estimate_planar_pose_batch", "views": args.views, "points_per_view": n_pts, "wall_s": t_s,
                      "views_per_s": args.views / t_s, "max_abs_err_vs_gt_first_200": float(err)}))
    # semi-DLT intrinsics: 400 views x 30x30 points, seeded on the device, distortion eliminated by variable projection
    from tests import helpers
    d, kgt, agt = helpers.semidlt_scene(400, rows=30, cols=30, noise=0.2, nr=2, seed=5)
    vs = [np.c_[d["X"][a:b], d["Y"][a:b], d["u"][a:b], d["v"][a:b]] for a, b in zip(d["off"][:-1], d["off"][1:])]
    opt = optim.IntrinsicsOptimOptions(core=optim.OptimOptions(compute_covariance=False), num_radial=2)
    for rep in range(2):
        t0 = time.perf_counter()
        r = optim.optimize_intrinsics_semidlt(vs, d["kappa0"], None, opt)
        t_sd = time.perf_counter() - t0
    print(json.dumps({"case": "optimize_intrinsics_semidlt (device seeds)", "views": 400, "points_per_view": 900, "wall_s": t_sd,
                      "solve_s": r.core.solve_seconds, "iterations": r.core.iterations, "success": r.core.success,
                      "K_err_px": float(np.abs(r.camera[:4] - kgt[:4]).max()), "alpha_err": float(np.abs(r.distortion - agt).max())}))


if __name__ == "__main__":
    main()
