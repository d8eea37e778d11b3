import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from calibration_amd import capi, optim, synth
from tests import helpers
sc = synth.scene_bundle(2000, 4, noise_px=0.2, seed=2024)
f = sc.flat
start = (f.intr.copy(), f.cam_pose.copy(), f.target_pose.copy())
o = helpers.options(optimize_intrinsics=1, compute_covariance=0)
with optim.ReprojHandle(f) as h:
    for rep in range(4):
        h.set_params(intr=start[0], cam_pose=start[1], target_pose=start[2])
        t0 = time.perf_counter(); s = h.solve(o); dt = time.perf_counter() - t0
        print(json.dumps({"rep": rep, "ms": dt * 1e3, "iters": s.iterations, "stats": h.solve_stats()}), flush=True)
