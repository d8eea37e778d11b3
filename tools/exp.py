#!/usr/bin/env python3
"""One-off measurement scripts of the build, as sub-commands: python tools/exp.py <name> [args...]   (python tools/exp.py list).

Each was written to answer ONE question about a kernel or the LM on an MI355X (the answers are in DESIGN.md and profiles/*_notebook.md);
they are kept so that a number can be re-measured.  Experiment knobs (CBA_MODEB_VARIANT, CBA_EVAL_ABLATE, ...) are read only by a
library built with -DCBA_EXPERIMENTS: `make -C calibration_amd/csrc EXPERIMENTS=1 LIBDIR=../lib_exp OBJDIR=_build_exp`; when
calibration_amd/lib_exp/libcalibba.so exists this driver selects it (CALIBBA_LIBRARY), otherwise the shipped library runs and
ignores those knobs."""
import json, os, statistics, sys, time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
_exp_lib = os.path.join(ROOT, "calibration_amd", "lib_exp", "libcalibba.so")
if os.path.exists(_exp_lib):
    os.environ.setdefault("CALIBBA_LIBRARY", _exp_lib)
import numpy as np

EXPERIMENTS = {}


def experiment(fn):
    EXPERIMENTS[fn.__name__] = fn
    return fn


@experiment
def abl(argv):
    from calibration_amd import optim
    from tests import synth
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    h = optim.ReprojHandle(sc.flat)
    os.environ["CBA_EVAL_VARIANT"], os.environ["CBA_EVAL_BLOCKED"] = "1", "1"
    res = {a: [] for a in (0, 1, 2)}
    for rnd in range(8):
        for a in (0, 1, 2):
            os.environ["CBA_EVAL_ABLATE"] = str(a)
            res[a].append(h.eval_timed(2, 20))
    for a, name in ((0, "full"), (1, "no arithmetic (loads + stores)"), (2, "no loads (arithmetic + stores)")):
        m = min(res[a]); print(f"{name:34s}: min {m:.4f} ms ({304e7/m/1e6:.0f} GB/s alg)  median {statistics.median(res[a]):.4f}")


@experiment
def abl22(argv):
    """Timing ablation of Mode A for the two-pose chain (P = 22): full kernel / no arithmetic / no loads (see the `abl` experiment for P = 16)."""
    from calibration_amd import optim
    from tests import synth
    sc = synth.scene_extrinsics(200, 8, rows=50, cols=100, spacing=0.008, noise_px=0.2, seed=137)
    h = optim.ReprojHandle(sc.flat)
    n = sc.flat.n_obs
    os.environ["CBA_EVAL_VARIANT"], os.environ["CBA_EVAL_BLOCKED"] = "1", "1"
    res = {a: [] for a in (0, 1, 2)}
    for rnd in range(8):
        for a in (0, 1, 2):
            os.environ["CBA_EVAL_ABLATE"] = str(a)
            res[a].append(h.eval_timed(2, 20))
    for a, name in ((0, "full"), (1, "no arithmetic (loads + stores)"), (2, "no loads (arithmetic + stores)")):
        m = min(res[a]); print(f"{name:34s}: min {m:.4f} ms ({400 * n / m / 1e6:.0f} GB/s alg)  median {statistics.median(res[a]):.4f}")


@experiment
def blk(argv):
    """k_eval variants timed on the SAME buffers (knobs re-read per timed call), interleaved rounds."""
    from calibration_amd import optim
    from tests import synth
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    cfgs = [tuple(int(x) for x in a.split(",")) for a in argv] or [(1, 1), (3, 1), (5, 1), (7, 1), (0, 1), (2, 1), (4, 1)]
    h = optim.ReprojHandle(sc.flat)
    res = {c: [] for c in cfgs}
    for rnd in range(8):
        for c in cfgs:
            os.environ["CBA_EVAL_VARIANT"], os.environ["CBA_EVAL_BLOCKED"] = str(c[0]), str(c[1])
            res[c].append(h.eval_timed(2, 20))
    for c in cfgs:
        m, md = min(res[c]), statistics.median(res[c])
        print(f"  variant {c[0]} blocked {c[1]}: min {m:.4f} ms ({304e7/m/1e6:.0f} GB/s)  median {md:.4f} ms ({304e7/md/1e6:.0f} GB/s)")


@experiment
def c1_lm(argv):
    """LM wall time of a C1-sized problem (20 views x 88 points) solved three times on one handle: with CBA_LM_GRAPH=1 the first
    solve pays the HIP-graph capture + instantiation of the three stages, the later ones replay them."""
    from calibration_amd import optim, capi
    from tests import synth
    sc = synth.scene_intrinsics(20, noise_px=0.2)
    init = (sc.flat.intr.copy(), sc.flat.view_pose.copy())
    o = capi.default_options(); o.compute_covariance = 0
    with optim.ReprojHandle(sc.flat) as h:
        for k in range(3):
            h.set_params(intr=init[0], view_pose=init[1])
            t0 = time.perf_counter(); s = h.solve(o); dt = time.perf_counter() - t0
            print(f"CBA_LM_GRAPH={os.environ.get('CBA_LM_GRAPH', 'default')} CBA_LM_RESIDENT={os.environ.get('CBA_LM_RESIDENT', 'default')} solve {k}: "
                  f"{dt*1e3:.2f} ms ({s.solve_seconds*1e3:.2f} in the engine), {s.iterations} iterations, {dt/s.iterations*1e6:.0f} us/iteration, "
                  f"cost {s.final_cost:.9e}, {bytes(s.report).split(b':')[0].decode()}")


@experiment
def c4_lm(argv):
    from calibration_amd import capi, optim
    from tests import synth
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


@experiment
def c5_fp32(argv):
    """BASELINE configs[4] study at full size: Scheimpflug intrinsics, 1000 views x 10 000 points, 0.2 px noise, fp32 kernels vs fp64.
    Writes one JSON object (Mode A / Mode B / LM time, row errors against fp64, end-state deviation against fp64 and ground truth)."""
    from calibration_amd import capi, optim
    from tests import synth
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


@experiment
def create(argv):
    """Phase timing of cba_reproj_create at C3 size (CBA_CREATE_TIMING=1 prints the phases to stderr)."""
    os.environ["CBA_CREATE_TIMING"] = "1"
    from calibration_amd import optim
    from tests import synth
    scale = float(argv[0]) if len(argv) > 0 else 1.0
    sc = synth.scene_extrinsics(int(4000 * scale), 8, rows=50, cols=100, spacing=0.008, noise_px=0.2, seed=137)
    for k in range(2):
        t0 = time.perf_counter(); h = optim.ReprojHandle(sc.flat); print(f"create {k}: {time.perf_counter() - t0:.3f} s for {sc.flat.n_obs} observations", flush=True); h.close()
    # the same from {X, Y, u, v} records read in place (cba_reproj_create_aos): what a binding to the reference's PlanarView pays
    f = sc.flat
    t0 = time.perf_counter()
    rec = np.empty((f.n_obs, 4)); rec[:, 0] = f.X; rec[:, 1] = f.Y; rec[:, 2] = f.u; rec[:, 3] = f.v
    records = [rec[a:b] for a, b in zip(f.blk_offset[:-1], f.blk_offset[1:])]
    print(f"(building the record arrays for this experiment: {time.perf_counter() - t0:.3f} s — the copy a caller with AoS data avoids the reverse of)", flush=True)
    for k in range(2):
        t0 = time.perf_counter(); h = optim.ReprojHandle(f, records=records); print(f"create from records {k}: {time.perf_counter() - t0:.3f} s", flush=True); h.close()


@experiment
def eval(argv):
    """Interleaved A/B timing of k_eval variants in ONE process (one handle per variant, variant chosen by
    CBA_EVAL_VARIANT at handle creation), N rounds; prints min / median ms per eval."""
    from calibration_amd import optim
    from tests import synth
    variants = [int(a) for a in argv] or [0, 1, 2, 3]
    shape = os.environ.get("EXP_SHAPE", "c2")  # c2 (P = 16), c5 (Scheimpflug, P = 18), c3q (8-camera rig / 4, P = 22)
    if shape == "c5":
        sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2, model=1, seed=5)
    elif shape == "c3q":
        sc = synth.scene_extrinsics_shard(4000, 0, 1000)
    else:
        sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    bytes_per = {"c2": 304, "c5": 336, "c3q": 400}[shape] * sc.flat.n_obs
    hs = {}
    for v in variants:
        os.environ["CBA_EVAL_VARIANT"] = str(v)
        hs[v] = optim.ReprojHandle(sc.flat)
        hs[v].eval_timed(3, 5)
    res = {v: [] for v in variants}
    for rnd in range(12):
        for v in variants:
            res[v].append(hs[v].eval_timed(1, 20))
    for v in variants:
        m, md = min(res[v]), statistics.median(res[v])
        print(f"{shape} variant {v}: min {m:.4f} ms ({bytes_per/m/1e6:.0f} GB/s)  median {md:.4f} ms ({bytes_per/md/1e6:.0f} GB/s)")


@experiment
def first_eval(argv):
    from calibration_amd import optim
    from tests import synth
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    with optim.ReprojHandle(sc.flat) as h:
        t0 = time.perf_counter(); h.eval(); t1 = time.perf_counter(); h.eval(); t2 = time.perf_counter()
        print(f"CBA_EVAL_CONTIGUOUS={os.environ.get('CBA_EVAL_CONTIGUOUS','default (1)')}: first eval {1e3*(t1-t0):.1f} ms, second {1e3*(t2-t1):.2f} ms, rate {304e7/h.eval_timed(2,20)/1e6:.0f} GB/s")


@experiment
def first_eval_c3(argv):
    """Cost of the physically contiguous Mode A output block at larger sizes: first / second evaluation time and rate of the 8-camera rig with
    <views> views (output = views x 8 x 5000 x 368 B)."""
    from calibration_amd import optim
    from tests import synth
    views = int(argv[0]) if len(argv) > 0 else 4000
    sc = synth.scene_extrinsics_shard(4000, 0, views)
    with optim.ReprojHandle(sc.flat) as h:
        t0 = time.perf_counter(); h.eval(); t1 = time.perf_counter(); h.eval(); t2 = time.perf_counter()
        n = sc.flat.n_obs
        print(f"{views} views ({n * 368 / 2**30:.1f} GiB out) CBA_EVAL_CONTIGUOUS={os.environ.get('CBA_EVAL_CONTIGUOUS', 'default (1)')}: first eval "
              f"{1e3 * (t1 - t0):.1f} ms, second {1e3 * (t2 - t1):.2f} ms, rate {400 * n / h.eval_timed(1, 5) / 1e6:.0f} GB/s", flush=True)


@experiment
def fp32_modeb(argv):
    from calibration_amd import optim
    from tests import synth
    for model in (0, 1):
        sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2, model=model, seed=5)
        with optim.ReprojHandle(sc.flat) as h:
            a = min(h.normal_eq_timed(2, 10) for _ in range(3))
            h.set_scalar(1)
            b = min(h.normal_eq_timed(2, 10) for _ in range(3))
        print(f"model {model}: Mode B fp64 {a:.3f} ms, fp32 rows {b:.3f} ms")


@experiment
def ld(argv):
    """Column-stride experiment for k_eval: (ld*8) mod 2 MiB forced to given byte offsets (CBA_LD_MOD),
    variant 5 (nt, 4 tiles/wave) and 0; interleaved rounds in one process."""
    from calibration_amd import optim
    from tests import synth
    mods = [int(a) for a in argv] or [-1, 0, 256, 512, 2048, 4096, 4352, 65792, 131328]
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    res = {}
    for variant in (0, 5):
        os.environ["CBA_EVAL_VARIANT"] = str(variant)
        for m in mods:
            if m < 0: os.environ.pop("CBA_LD_MOD", None)
            else: os.environ["CBA_LD_MOD"] = str(m)
            with optim.ReprojHandle(sc.flat) as h:
                h.eval_timed(3, 5)
                t = [h.eval_timed(1, 20) for _ in range(5)]
            res[(variant, m)] = min(t)
            print(f"variant {variant} ld_mod {m:7d}: min {min(t):.4f} ms ({304e7/min(t)/1e6:.0f} GB/s) median {statistics.median(t):.4f}", flush=True)


@experiment
def lm(argv):
    """LM wall clock of the BASELINE shapes on one MI355X, with the exchange statistics of the solve.
    usage: python tools/exp.py lm [c2] [c3] [c3q] [c5] [c1h]   (CBA_LM_SPECULATE=0 for the two-exchange sequence)"""
    from calibration_amd import capi, optim
    from tests import synth
    from tests import helpers

    which = argv or ["c2", "c3q"]
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


@experiment
def membw(argv):
    import torch, time
    n = 3_000_000_000 // 8
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    def t(fn, iters=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    ms = t(lambda: x.fill_(1.5)); print(f"fill  (write only) {n*8/ms/1e6:.0f} GB/s  {ms:.3f} ms")
    ms = t(lambda: x.zero_()); print(f"zero  (write only) {n*8/ms/1e6:.0f} GB/s  {ms:.3f} ms")
    ms = t(lambda: y.copy_(x)); print(f"copy  (r+w)       {2*n*8/ms/1e6:.0f} GB/s  {ms:.3f} ms")
    ms = t(lambda: x.sum()); print(f"sum   (read only)  {n*8/ms/1e6:.0f} GB/s  {ms:.3f} ms")
    ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"add   (r+w)        {2*n*8/ms/1e6:.0f} GB/s  {ms:.3f} ms")


@experiment
def modeb(argv):
    """Mode B (per-block normal equations) timing of the BASELINE shapes on one MI355X: ms per pass and the same per observation.
    usage: python tools/exp.py modeb [c2] [c5] [c3q] [c4]   (c3q = C3 at a quarter of the views)"""
    from calibration_amd import capi, optim
    from tests import synth

    which = argv or ["c2", "c5", "c3q"]
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


@experiment
def modeb_warm(argv):
    from calibration_amd import optim
    from tests import synth
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    with optim.ReprojHandle(sc.flat) as h:
        for w, n in ((2, 10), (2, 10), (50, 50), (200, 50), (2, 10), (500, 100), (2, 10)):
            print(f"warmup {w:4d} iters {n:4d}: {h.normal_eq_timed(w, n):.4f} ms", flush=True)
        h.eval_timed(3, 50)
        print(f"after 53 Mode A launches, warmup 2 iters 10: {h.normal_eq_timed(2, 10):.4f} ms")
        print(f"again: {h.normal_eq_timed(2, 10):.4f} ms")


@experiment
def oneshot(argv):
    """Wall time of the ONE-SHOT entry point (what the reference's pipeline calls): cba_optimize_intrinsics on a C1-sized problem
    (20 views x 88 points), repeated — handle creation + solve + covariance + destruction per call."""
    from calibration_amd import optim
    from tests import synth
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


@experiment
def placement(argv):
    """Does the position of the Mode A output buffer in the device heap change the write rate?  One process per pre-allocation size:
    `python tools/exp.py placement <GiB held before the handle is created>` -> ms per k_eval launch at C2."""
    import torch
    from calibration_amd import optim
    from tests import synth
    pre = float(argv[0]) if len(argv) > 0 else 0.0
    hold = torch.empty(int(pre * (1 << 30)), dtype=torch.uint8, device="cuda") if pre > 0 else None
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    with optim.ReprojHandle(sc.flat) as h:
        h.eval_timed(3, 5)
        t = [h.eval_timed(1, 20) for _ in range(12)]
    print(f"pre-allocated {pre:5.1f} GiB: min {min(t):.4f} ms ({304e7 / min(t) / 1e6:.0f} GB/s)  median {statistics.median(t):.4f} ms", flush=True)


@experiment
def placement2(argv):
    """k_eval rate of several handles of the same C2 problem created one after the other in ONE process: (a) all kept alive, (b) each
    released before the next is created."""
    from calibration_amd import optim
    from tests import synth
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    def rate(h):
        h.eval_timed(3, 5)
        return min(h.eval_timed(1, 20) for _ in range(8))
    alive = []
    for k in range(6):
        h = optim.ReprojHandle(sc.flat); alive.append(h)
        print(f"alive    #{k}: {rate(h):.4f} ms ({304e7 / rate(h) / 1e6:.0f} GB/s)", flush=True)
    print("again, in creation order: " + " ".join(f"{304e7 / rate(h) / 1e6:.0f}" for h in alive), flush=True)
    for h in alive: h.close()
    for k in range(6):
        with optim.ReprojHandle(sc.flat) as h:
            print(f"released #{k}: {rate(h):.4f} ms ({304e7 / rate(h) / 1e6:.0f} GB/s)", flush=True)


@experiment
def placement_modeb(argv):
    """Mode B rate of several handles of the same problem created one after the other in ONE process (kept alive): does the placement of a
    handle's buffers matter for the compute-bound kernel too?"""
    from calibration_amd import optim
    from tests import synth
    which = argv[0] if len(argv) > 0 else "c3q"
    sc = synth.scene_extrinsics_shard(4000, 0, 1000) if which == "c3q" else synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    alive = []
    for k in range(6):
        h = optim.ReprojHandle(sc.flat); alive.append(h)
        print(f"{which} handle #{k}: Mode B {min(h.normal_eq_timed(2, 10) for _ in range(3)):.4f} ms", flush=True)
    print("again: " + " ".join(f"{min(h.normal_eq_timed(2, 10) for _ in range(3)):.4f}" for h in alive), flush=True)


@experiment
def resident(argv):
    """LM wall time of small problems through the two forms of the iteration (cba_reproj_set_lm_mode): 0 = host-driven (every stage
    a kernel launch), 2 = resident (the whole solve in one single-workgroup kernel).  Where the curves cross is the default
    size limit of the automatic mode (CBA_LM_RESIDENT_MAX_OBS)."""
    from calibration_amd import optim, capi
    from tests import synth

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


@experiment
def resident_profile(argv):
    """Per-phase times of the resident LM kernel (CBA_LM_RESIDENT_PROFILE=1, printed by the library on stderr) for one small problem
    of each chain."""
    os.environ["CBA_LM_RESIDENT_PROFILE"] = "1"
    os.environ["CBA_LM_RESIDENT"] = "2"
    from calibration_amd import optim, capi
    from tests import synth
    o = capi.default_options(); o.compute_covariance = 0
    for name, sc in (("intr 20 x 88", synth.scene_intrinsics(20, noise_px=0.2)), ("ext 10 x 2 x 88", synth.scene_extrinsics(10, 2, noise_px=0.2)),
                     ("bundle 25 x 88", synth.scene_bundle(25, 1, noise_px=0.2))):
        with optim.ReprojHandle(sc.flat) as h:
            s = h.solve(o)
        print(f"{name}: {s.iterations} iterations, {s.solve_seconds*1e3:.2f} ms", file=sys.stderr, flush=True)


@experiment
def stages(argv):
    """Where a C1-sized calibration call spends its time: handle creation / solve / covariance / destruction, through the handle API."""
    import copy
    from calibration_amd import optim, capi
    from tests import synth
    sc = synth.scene_intrinsics(20, noise_px=0.2)
    o = capi.default_options()
    for k in range(5):
        f = copy.deepcopy(sc.flat)
        t = [time.perf_counter()]
        h = optim.ReprojHandle(f); t.append(time.perf_counter())
        s = h.solve(o); t.append(time.perf_counter())
        cov = h.covariance(o); t.append(time.perf_counter())
        h.close(); t.append(time.perf_counter())
        d = np.diff(t) * 1e3
        print(f"call {k}: create {d[0]:.2f}  solve {d[1]:.2f}  covariance {d[2]:.2f}  destroy {d[3]:.2f}  total {sum(d):.2f} ms")


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] in ("list", "-h", "--help") or sys.argv[1] not in EXPERIMENTS:
        print(__doc__)
        for k, fn in EXPERIMENTS.items():
            print(f"  {k:20s} {(fn.__doc__ or '').strip().splitlines()[0] if fn.__doc__ else ''}")
        raise SystemExit(0 if len(sys.argv) >= 2 and sys.argv[1] == "list" else 2)
    EXPERIMENTS[sys.argv[1]](sys.argv[2:])
