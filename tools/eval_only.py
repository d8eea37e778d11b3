#!/usr/bin/env python3
"""Mode A (k_eval) alone on one BASELINE shape, for profiler passes: creates the handle, runs W + K evaluations, prints ms per launch.
usage: python tools/eval_only.py <c2|c3|c3q|c4|c5> [iterations]   (under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: one counter per pass)"""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from calibration_amd import capi, optim
from tests import synth

w = sys.argv[1] if len(sys.argv) > 1 else "c2"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if w == "c2":
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.002, noise_px=0.2)
elif w == "c5":
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.002, noise_px=0.2, model=capi.CAMERA_SCHEIMPFLUG, seed=5)
elif w == "c3":
    sc = synth.scene_extrinsics_shard(4000, 0, 4000)
elif w == "c3q":
    sc = synth.scene_extrinsics_shard(4000, 0, 1000)
elif w == "c4":
    sc = synth.scene_bundle(2000, 4, seed=2024, noise_px=0.2, distortion=True)
else:
    raise SystemExit("unknown shape " + w)
with optim.ReprojHandle(sc.flat) as h:
    P = h.local_columns
    h.eval_timed(1, 1)
    ms = h.eval_timed(0, k)
    B = 8 * (4 + 2 + 2 * P)
    print(json.dumps({"shape": w, "n_obs": int(sc.flat.n_obs), "tangent_columns": P, "ms_per_launch": ms, "algorithmic_bytes_per_launch": B * int(sc.flat.n_obs),
                      "algorithmic_GBs": B * sc.flat.n_obs / (ms * 1e-3) / 1e9, "frac_of_8TBs": B * sc.flat.n_obs / (ms * 1e-3) / 8e12,
                      "segments": os.environ.get("CBA_EVAL_SEGMENTS", "1")}), flush=True)
