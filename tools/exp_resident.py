#!/usr/bin/env python3
"""LM wall time of small problems through the two forms of the iteration (cba_reproj_set_lm_mode): 0 = host-driven (every stage
a kernel launch), 2 = resident (the whole solve in one single-workgroup kernel).  Where the curves cross is the default
size limit of the automatic mode (CBA_LM_RESIDENT_MAX_OBS)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from calibration_amd import optim, synth, capi

CASES = [("intr", 10, 8, 11), ("intr", 20, 8, 11), ("intr", 50, 8, 11), ("intr", 20, 15, 15), ("intr", 20, 20, 20), ("intr", 50, 20, 20),
         ("intr", 100, 20, 20), ("ext", 10, 8, 11), ("bundle", 25, 8, 11), ("ext", 4, 5, 5), ("ext", 8, 6, 6), ("ext", 6, 8, 11), ("bundle", 10, 6, 6),
         ("bundle", 6, 8, 11), ("bundle", 12, 8, 11), ("intr", 30, 8, 11), ("intr", 10, 12, 12), ("intr", 6, 20, 20)]
o = capi.default_options(); o.compute_covariance = 0
print(f"{'case':28s} {'obs':>7s} {'iters':>5s} {'host ms':>9s} {'resident ms':>11s} {'us/iter host':>12s} {'us/iter res':>11s}")
for kind, nv, rows, cols in CASES:
    if kind == "intr": sc = synth.scene_intrinsics(nv, rows=rows, cols=cols, noise_px=0.2)
    elif kind == "ext": sc = synth.scene_extrinsics(nv, 2, rows=rows, cols=cols, noise_px=0.2)
    else: sc = synth.scene_bundle(nv, 1, rows=rows, cols=cols, noise_px=0.2)
    f = sc.flat
    init = [None if x is None else x.copy() for x in (f.intr, f.cam_pose, f.view_pose, f.target_pose)]
    res = {}
    with optim.ReprojHandle(f) as h:
        for mode in (0, 2):
            h.set_lm_mode(mode)
            best = None
            for rep in range(4):
                h.set_params(*init)
                t0 = time.perf_counter(); s = h.solve(o); dt = time.perf_counter() - t0
                if rep and (best is None or dt < best[0]): best = (dt, s.iterations, s.final_cost)
            res[mode] = best
        n = h.n_obs
    assert res[0][1] == res[2][1] and abs(res[0][2] - res[2][2]) <= 1e-9 * abs(res[0][2]), (res, kind)
    it = res[0][1]
    print(f"{kind + f' {nv} x {rows}x{cols}':28s} {n:7d} {it:5d} {res[0][0]*1e3:9.3f} {res[2][0]*1e3:11.3f} {res[0][0]/it*1e6:12.1f} {res[2][0]/it*1e6:11.1f}")
