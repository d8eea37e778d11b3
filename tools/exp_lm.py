#!/usr/bin/env python3
"""LM wall clock of the BASELINE shapes on one MI355X, with the exchange statistics of the solve.
usage: python tools/exp_lm.py [c2] [c3] [c3q] [c5] [c1h]   (CBA_LM_SPECULATE=0 for the two-exchange sequence)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from calibration_amd import capi, optim, synth
from tests import helpers

which = sys.argv[1:] or ["c2", "c3q"]
capi.load_library()
for w in which:
    if w == "c2":
        sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    elif w == "c5":
        sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2, model=capi.CAMERA_SCHEIMPFLUG, seed=5)
    elif w == "c3":
        sc = synth.scene_extrinsics_shard(4000, 0, 4000)
    elif w == "c3q":
        sc = synth.scene_extrinsics_shard(1000, 0, 1000)
    elif w == "c3e":  # the share of ONE rank when C3 is split over 8 GPUs (500 of the 4000 views): what a rank computes per step, without the exchange
        sc = synth.scene_extrinsics_shard(4000, 0, 500)
    elif w == "c3r2":  # ... over 2 GPUs (2000 views)
        sc = synth.scene_extrinsics_shard(4000, 0, 2000)
    elif w == "c3r4":  # ... over 4 GPUs (1000 views)
        sc = synth.scene_extrinsics_shard(4000, 0, 1000)
    elif w == "c1h":
        os.environ["CBA_LM_RESIDENT"] = "0"
        sc = synth.scene_intrinsics(20, noise_px=0.2)
    else:
        raise SystemExit("unknown " + w)
    f = sc.flat
    start = (f.intr.copy(), None if f.cam_pose is None else f.cam_pose.copy(), f.view_pose.copy())
    with optim.ReprojHandle(f) as h:
        walls = []
        for rep in range(3):
            h.set_params(intr=start[0], cam_pose=start[1], view_pose=start[2])
            t = time.perf_counter()
            s = h.solve(helpers.options(compute_covariance=0, verbose=int(os.environ.get('EXP_VERBOSE', '0')) if rep == 0 else 0))
            walls.append(time.perf_counter() - t)
        print(json.dumps({"shape": w, "n_obs": int(f.n_obs), "lm_wall_ms": [round(x * 1e3, 3) for x in walls], "iterations": int(s.iterations),
                          "accepted": int(s.successful_steps), "speculate": os.environ.get("CBA_LM_SPECULATE", "1"), "stats": h.solve_stats(),
                          "mode_b_ms": h.normal_eq_timed(1, 5), "report": s.report.decode()}), flush=True)
