#!/usr/bin/env python3
"""Measures every BASELINE.json config shape on one MI355X (run through gpurun); writes JSON lines to
gpurun_out/configs.jsonl.  Not the driver's bench (that is bench.py = configs[1]); this is the evidence for the
other rows of SURVEY.md §8(d).  Scale can be reduced with --scale (fraction of the full view counts)."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from calibration_amd import capi, optim
from tests import synth
from calibration_amd.geometry import pose_from_matrix, pose_to_matrix, rotation_angle
from tests import helpers

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--only", default="")
args = ap.parse_args()
out_path = os.path.join(ROOT, "gpurun_out", "configs.jsonl")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
lib = capi.load_library()
orc = helpers.load_oracle() if os.path.exists(helpers.ORACLE_SO) else None


def emit(rec):
    print(json.dumps(rec), flush=True)
    with open(out_path, "a") as f:
        f.write(json.dumps(rec) + "\n")


def run_reproj(name, scene, okw, eval_iters=10, oracle_compare=False):
    flat = scene.flat
    t0 = time.time()
    h = optim.ReprojHandle(flat)
    t_create = time.time() - t0
    P = h.local_columns
    B = 8 * (4 + 2 + 2 * P)
    ms = min(h.eval_timed(2, eval_iters) for _ in range(3))
    init = (flat.intr.copy(), None if flat.cam_pose is None else flat.cam_pose.copy(),
            None if flat.view_pose is None else flat.view_pose.copy(), None if flat.target_pose is None else flat.target_pose.copy())
    o = helpers.options(compute_covariance=0, **okw)
    t1 = time.perf_counter()
    s = h.solve(o)
    lm_s = time.perf_counter() - t1
    rec = {"config": name, "n_obs": flat.n_obs, "n_blocks": flat.n_blocks, "n_views": flat.n_views, "n_cams": flat.n_cams,
           "tangent_columns": P, "mode_a_ms": ms, "evals_per_s": flat.n_obs / (ms * 1e-3), "bytes_per_eval": B,
           "hbm_GBs": B * flat.n_obs / (ms * 1e-3) / 1e9, "hbm_frac_of_8TBs": B * flat.n_obs / (ms * 1e-3) / 8e12,
           "lm_wall_s": lm_s, "lm_iterations": int(s.iterations), "lm_success": bool(s.success), "lm_final_cost": float(s.final_cost),
           "intr_err_max": float(np.abs(flat.intr - scene.gt_intr).max()), "handle_create_s": t_create}
    if oracle_compare and orc is not None:
        import copy
        ref = copy.deepcopy(flat)
        ref.intr[...], = (init[0],)
        if init[1] is not None: ref.cam_pose[...] = init[1]
        if init[2] is not None: ref.view_pose[...] = init[2]
        if init[3] is not None: ref.target_pose[...] = init[3]
        t2 = time.perf_counter()
        so = helpers.oracle_solve(orc, ref, o, threads=16)
        rec.update({"oracle_lm_wall_s": time.perf_counter() - t2, "oracle_iterations": int(so.iterations),
                    "param_diff_vs_oracle": helpers.param_diff(ref, flat)})
    h.close()
    emit(rec)


sel = set(args.only.split(",")) if args.only else None
sc = args.scale
if not sel or "C1" in sel:
    run_reproj("C1 pinhole intrinsics 20 views x 88 pts", synth.scene_intrinsics(20, noise_px=0.2), {}, oracle_compare=True)
if not sel or "C2" in sel:
    run_reproj("C2 pinhole+BC intrinsics 1000 x 10000", synth.scene_intrinsics(int(1000 * sc), rows=100, cols=100, spacing=0.008, noise_px=0.2), {})
if not sel or "C5" in sel:
    run_reproj("C5 Scheimpflug intrinsics 1000 x 10000 (fp64)", synth.scene_intrinsics(int(1000 * sc), rows=100, cols=100, spacing=0.008, noise_px=0.2, model=1, seed=5), {})
if not sel or "C4" in sel:
    run_reproj("C4 hand-eye bundle 2000 poses x 4 cams x 88 pts", synth.scene_bundle(int(2000 * sc), 4, noise_px=0.2, seed=2024), dict(optimize_intrinsics=1))
    # AX=XB over the same number of robot poses
    n = int(2000 * sc)
    bTg, cTt, X_gt, X0 = helpers.handeye_scene(n, seed=2024, noise_rot_deg=0.1, noise_trans=0.001)
    pb = np.stack([pose_from_matrix(T) for T in bTg]); pc = np.stack([pose_from_matrix(T) for T in cTt]); x = pose_from_matrix(X0)
    o = helpers.options(); s = capi.CbaSummary(); cov = np.zeros((7, 7))
    t1 = time.perf_counter()
    capi.check(lib, lib.cba_optimize_handeye(n, capi.dptr(pb), capi.dptr(pc), capi.dptr(x), C.byref(o), C.byref(s), capi.dptr(cov)))
    wall = time.perf_counter() - t1
    X = pose_to_matrix(x)
    emit({"config": f"C4 AX=XB {n} poses", "report": s.report.decode(), "wall_s": wall, "iterations": int(s.iterations),
          "rot_err_deg": float(np.rad2deg(rotation_angle(X[:3, :3].T @ X_gt[:3, :3]))), "trans_err": float(np.linalg.norm(X[:3, 3] - X_gt[:3, 3]))})
if not sel or "C3" in sel:
    t0 = time.time()
    scene = synth.scene_extrinsics(int(4000 * sc), 8, rows=50, cols=100, spacing=0.008, noise_px=0.2, seed=137)
    print("C3 scene generated in %.1f s" % (time.time() - t0), flush=True)
    run_reproj("C3 8-camera extrinsics 4000 views x 5000 pts/view/cam", scene, {}, eval_iters=3)
