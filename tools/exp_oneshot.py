#!/usr/bin/env python3
"""Wall time of the ONE-SHOT entry point (what the reference's pipeline calls): cba_optimize_intrinsics on a C1-sized problem
(20 views x 88 points), repeated — handle creation + solve + covariance + destruction per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from calibration_amd import optim, synth
from calibration_amd.geometry import pose_to_matrix
sc = synth.scene_intrinsics(20, noise_px=0.2)
f = sc.flat
views = [np.c_[f.X[a:b], f.Y[a:b], f.u[a:b], f.v[a:b]] for a, b in zip(f.blk_offset[:-1], f.blk_offset[1:])]
poses = [pose_to_matrix(p) for p in f.view_pose]
for k in range(5):
    t0 = time.perf_counter()
    r = optim.optimize_intrinsics(views, f.intr.reshape(-1).copy(), poses)
    dt = time.perf_counter() - t0
    print(f"call {k}: {dt*1e3:.2f} ms total, solve {r.core.solve_seconds*1e3:.2f} ms, {r.core.iterations} iterations, success {r.core.success}")
