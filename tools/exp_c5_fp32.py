#!/usr/bin/env python3
"""BASELINE configs[4] study at full size: Scheimpflug intrinsics, 1000 views x 10 000 points, 0.2 px noise, fp32 kernels vs fp64.
Writes one JSON object (Mode A / Mode B / LM time, row errors against fp64, end-state deviation against fp64 and ground truth)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from calibration_amd import capi, optim, synth
from tests import helpers

res = {}
for scalar in (0, 1):
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2, model=capi.CAMERA_SCHEIMPFLUG, seed=5)
    start = (sc.flat.intr.copy(), sc.flat.view_pose.copy())
    with optim.ReprojHandle(sc.flat) as h:
        h.set_scalar(scalar)
        ms_a = min(h.eval_timed(2, 10) for _ in range(3))
        ms_b = min(h.normal_eq_timed(2, 10) for _ in range(3))
        h.eval()
        r, J = (h.eval_fetch_f32() if scalar else h.eval_fetch_blocks(0, 8))
        r, J = np.asarray(r[:160000], dtype=np.float64), np.asarray(J[:160000], dtype=np.float64)
        walls = []
        for _ in range(3):
            h.set_params(intr=start[0], view_pose=start[1])
            t = time.perf_counter()
            s = h.solve(helpers.options(compute_covariance=0))
            walls.append(time.perf_counter() - t)
        cs = h.covariance_shared(helpers.options()) if not scalar else None
    P = 18
    res[scalar] = dict(mode_a_ms=ms_a, evals_per_s=sc.flat.n_obs / (ms_a * 1e-3), bytes_per_eval=(4 if scalar else 8) * (4 + 2 + 2 * P),
                       hbm_GBs=(4 if scalar else 8) * (4 + 2 + 2 * P) * sc.flat.n_obs / (ms_a * 1e-3) / 1e9, mode_b_ms=ms_b, lm_wall_ms=min(walls) * 1e3,
                       lm_iterations=int(s.iterations), final_cost=float(s.final_cost), intr=sc.flat.intr.reshape(-1).tolist(), r=r, J=J,
                       sigma=None if cs is None else np.sqrt(np.diag(cs))[:12].tolist(), gt=sc.gt_intr.reshape(-1).tolist())
a, b = res[0], res[1]
out = {"workload": "Scheimpflug intrinsics, 1000 views x 10000 pts, 0.2 px noise (BASELINE configs[4])",
       "fp64": {k: a[k] for k in ("mode_a_ms", "evals_per_s", "bytes_per_eval", "hbm_GBs", "mode_b_ms", "lm_wall_ms", "lm_iterations", "final_cost")},
       "fp32": {k: b[k] for k in ("mode_a_ms", "evals_per_s", "bytes_per_eval", "hbm_GBs", "mode_b_ms", "lm_wall_ms", "lm_iterations", "final_cost")},
       "residual_abs_err_max_px": float(np.abs(b["r"] - a["r"]).max()),
       "jacobian_rel_err_max": float((np.abs(b["J"] - a["J"]) / np.maximum(1.0, np.abs(a["J"]))).max()),
       "jacobian_rel_err_rms": float(np.sqrt(np.mean(((b["J"] - a["J"]) / np.maximum(1.0, np.abs(a["J"]))) ** 2))),
       "intr_names": ["fx", "fy", "cx", "cy", "skew", "k1", "k2", "k3", "p1", "p2", "tau_x", "tau_y"],
       "fp32_minus_fp64": (np.array(b["intr"]) - np.array(a["intr"])).tolist(),
       "fp64_minus_ground_truth": (np.array(a["intr"]) - np.array(a["gt"])).tolist(), "fp64_sigma": a["sigma"],
       "cost_rel_diff": abs(b["final_cost"] - a["final_cost"]) / a["final_cost"]}
print(json.dumps(out))
