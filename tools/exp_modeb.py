#!/usr/bin/env python3
"""Mode B (per-block normal equations) timing of the BASELINE shapes on one MI355X: ms per pass and the same per observation.
usage: python tools/exp_modeb.py [c2] [c5] [c3q] [c4]   (c3q = C3 at a quarter of the views)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from calibration_amd import capi, optim, synth

which = sys.argv[1:] or ["c2", "c5", "c3q"]
lib = capi.load_library()
tag = os.environ.get("EXP_TAG", "")
for w in which:
    if w == "c2":
        sc = synth.scene_intrinsics(1000, 100, 100, 0.002, seed=11, noise_px=0.2)
    elif w == "c5":
        sc = synth.scene_intrinsics(1000, 100, 100, 0.002, model=capi.CAMERA_SCHEIMPFLUG, seed=11, noise_px=0.2)
    elif w == "c3q":
        sc = synth.scene_extrinsics(1000, 8, 50, 100, 0.004, seed=3, noise_px=0.2)
    elif w == "c3qs":
        sc = synth.scene_extrinsics(500, 8, 50, 100, 0.004, model=capi.CAMERA_SCHEIMPFLUG, seed=3, noise_px=0.2)
    elif w == "c4":
        sc = synth.scene_bundle(2000, 4, seed=5, noise_px=0.2, distortion=True)
    else:
        raise SystemExit("unknown " + w)
    with optim.ReprojHandle(sc.flat) as h:
        ms = min(h.normal_eq_timed(2, 10) for _ in range(3))
        n = sc.flat.n_obs
        print(json.dumps({"tag": tag, "shape": w, "n_obs": int(n), "mode_b_ms": ms, "ns_per_obs": ms * 1e6 / n,
                          "split": os.environ.get("CBA_MODEB_SPLIT", "1"), "lib": os.environ.get("CALIBBA_LIBRARY", "default")}), flush=True)
