#!/usr/bin/env python3
"""Randomised parity sweep of the HIP engine against the CPU oracle: random chain / camera model / sizes / noise / stage switches /
loss, LM to epsilon = 1e-12, compared on termination, iterations, cost and parameters.  Every case that misses the bar is
CLASSIFIED (tests/helpers.py solution_gap_report / gap_category): it is benign only if the two solvers terminated the same way,
nothing moved that Ceres holds constant, the costs differ by no more than the displacement explains, and either nearly all
(>= 95 %) of the scaled parameter difference lies in the three weakest eigen-directions of an ill-conditioned Hessian
("weak-direction"), or the quadratic model prices the whole displacement below 4 eps cost - less than the solvers' own function
tolerance resolves ("stopping-resolution") - or both solvers crept for >= 50 iterations on a slowly converging problem and stopped by the function
tolerance at costs equal to 1e-8 ("slow-convergence").  Anything else is an UNEXPLAINED disagreement: the sweep prints it and
exits 1.  Not part of the test suite (a search, run with spare GPU
time; the cases it found are pinned in tests/test_gpu_parity.py); prints one JSON summary line.
usage: python tools/fuzz_gpu.py [n_cases] [seed]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calibration_amd import capi, optim
from tests import synth
from tests import helpers

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = helpers.load_oracle()
hm = helpers.load_hostmath()
worst, bad, unexplained, t0 = 0.0, [], [], time.time()
for i in range(n_cases):
    kind = ["intr", "ext", "bundle"][int(rng.integers(0, 3))]
    model = int(rng.integers(0, 2))
    seed = int(rng.integers(1, 1 << 20))
    noise = float(rng.choice([0.0, 0.1, 0.5]))
    okw = dict(huber_delta=float(rng.choice([1.0, -1.0, 0.3, 3.0])), optimize_skew=int(rng.integers(0, 2)))
    if kind != "intr":
        okw.update(optimize_intrinsics=int(rng.integers(0, 2)), optimize_extrinsics=int(rng.integers(0, 2)))
    if kind == "bundle":
        okw.update(optimize_target_pose=int(rng.integers(0, 2)))
    nv, nc = int(rng.integers(4, 9)), int(rng.integers(1, 4))
    rows, cols = int(rng.integers(4, 10)), int(rng.integers(5, 12))
    mk = {"intr": lambda: synth.scene_intrinsics(nv, rows=rows, cols=cols, spacing=0.08, model=model, noise_px=noise, seed=seed),
          "ext": lambda: synth.scene_extrinsics(nv, max(2, nc), rows=rows, cols=cols, spacing=0.08, model=model, noise_px=noise, seed=seed),
          "bundle": lambda: synth.scene_bundle(nv + 4, nc, rows=rows, cols=cols, spacing=0.04, model=model, noise_px=noise, seed=seed)}[kind]
    a, b = mk(), mk()
    o = helpers.options(epsilon=1e-12, **okw)
    sa = helpers.oracle_solve(orc, a.flat, o)
    with optim.ReprojHandle(b.flat) as h:
        sb = h.solve(o)
    pd = helpers.param_diff(a.flat, b.flat)
    # Scheimpflug / free skew: near-degenerate directions amplify rounding between two correct solvers
    tol = 1e-5 if model == 1 else (5e-8 if okw.get("optimize_skew") else 2e-9)
    rec = dict(i=i, kind=kind, model=model, seed=seed, noise=noise, okw=okw, nv=nv, nc=nc, grid=[rows, cols], term=[int(sa.termination), int(sb.termination)],
               iters=[int(sa.iterations), int(sb.iterations)], cost=[float(sa.final_cost), float(sb.final_cost)], param_diff=pd)
    ok = sa.termination == sb.termination and abs(sa.iterations - sb.iterations) <= 2 and pd <= tol and \
        abs(sa.final_cost - sb.final_cost) <= 1e-8 * max(1.0, sa.final_cost) + 1e-14
    if not ok:
        rep = helpers.solution_gap_report(orc, hm, a.flat, b.flat, o)
        rec["gap"] = {k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in rep.items()}
        # the same termination - or the iteration cap between them: one side converges a few steps before max_iterations, the other
        # is cut off at it (a 1000-iteration creep along a flat valley; the categories below still have to explain the gap)
        cap = int(o.max_iterations)
        at_cap = {int(sa.termination), int(sb.termination)} == {capi.TERM_CONVERGENCE, capi.TERM_NO_CONVERGENCE} and \
            max(sa.iterations, sb.iterations) >= cap and min(sa.iterations, sb.iterations) >= 0.95 * cap
        rec["category"] = helpers.gap_category(rep, sa.final_cost, sb.final_cost, (int(sa.iterations), int(sb.iterations))) \
            if (sa.termination == sb.termination or at_cap) else "unexplained"
        if at_cap:
            rec["at_iteration_cap"] = True
        bad.append(rec)
        if rec["category"] == "unexplained":
            unexplained.append(rec)
    if model == 0 and not okw.get("optimize_skew"):
        worst = max(worst, pd)
print(json.dumps(dict(cases=n_cases, above_the_bar=len(bad), weak_direction=sum(r["category"] == "weak-direction" for r in bad),
                      stopping_resolution=sum(r["category"] == "stopping-resolution" for r in bad),
                      slow_convergence=sum(r["category"] == "slow-convergence" for r in bad), unexplained=len(unexplained),
                      worst_param_diff_pinhole_noskew=worst, seconds=time.time() - t0, unexplained_cases=unexplained[:10], benign_cases=bad[:10])))
sys.exit(1 if unexplained else 0)
