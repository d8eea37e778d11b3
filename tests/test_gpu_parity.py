"""GPU tier: the HIP path, called through the C ABI, against the CPU oracle on identical inputs.

Tolerances (fp64):
  residuals            |dr| <= 1e-9 px absolute (values ~1e1..1e3 px)
  Jacobian entries     |dJ| <= 1e-9 * max(1, |J|)
  block normal eq.     relative 1e-10 of the block's largest entry (different summation order)
  LM final parameters  max|dp|/max(|p|,1) <= 1e-9 for pinhole + Brown-Conrady (north-star bar) when
                       both solvers run with epsilon = 1e-12, so the bar measures arithmetic parity and
                       not where each solver happened to stop (at the reference's default epsilon =
                       1e-9 the last accepted step is itself ~1e-9 relative; that case is held to 1e-7);
                       1e-6 for the Scheimpflug model, whose tau/principal-point near-degeneracy
                       (condition number ~1e8) amplifies rounding between two correct solvers.
"""
import copy
import ctypes as C
import os

import numpy as np
import pytest

from calibration_amd import capi, optim
from tests import synth
from tests import helpers
from tests.helpers import options

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["host", "resident"])
def lm_mode(request, monkeypatch):
    """Every test of this module runs twice: with the host-driven LM iteration (backend_hip.hip + lm_core.hpp) and with the
    resident single-launch kernel (resident_lm.hip) wherever that kernel can take the problem.  The handle reads the
    variable when it is created."""
    monkeypatch.setenv("CBA_LM_RESIDENT", "0" if request.param == "host" else "2")
    return request.param

SCENES = {
    "intr": lambda m, **k: synth.scene_intrinsics(7, model=m, **k),
    "ext": lambda m, **k: synth.scene_extrinsics(5, 3, model=m, **k),
    "bundle": lambda m, **k: synth.scene_bundle(9, 2, model=m, distortion=True, **k),
}


def _perturb_intr(sc, seed=1):
    sc.flat.intr[...] = sc.gt_intr * (1 + 0.01 * np.random.default_rng(seed).uniform(-1, 1, sc.gt_intr.shape))


@pytest.mark.parametrize("model", [0, 1])
@pytest.mark.parametrize("kind", ["intr", "ext", "bundle"])
def test_mode_a_residual_and_jacobian(gpu_lib, oracle, kind, model):
    sc = SCENES[kind](model)
    _perturb_intr(sc)
    r0, J0 = helpers.oracle_eval(oracle, sc.flat)
    with optim.ReprojHandle(sc.flat) as h:
        h.eval()
        r1, J1 = h.eval_fetch()
    assert np.abs(r0 - r1).max() <= 1e-9
    assert (np.abs(J0 - J1) / np.maximum(1.0, np.abs(J0))).max() <= 1e-9


def test_mode_a_odd_and_tiny_views(gpu_lib, oracle):
    """ragged input: views of 1, 2, 3, 127, 128, 129, 257 points (tile / padding edges)."""
    rng = np.random.default_rng(3)
    sc = synth.scene_intrinsics(7, rows=17, cols=17)
    f = sc.flat
    keep = [1, 2, 3, 127, 128, 129, 257]
    views = []
    for b, n in enumerate(keep):
        lo = f.blk_offset[b]
        views.append(np.stack([f.X[lo:lo + n], f.Y[lo:lo + n], f.u[lo:lo + n], f.v[lo:lo + n]], axis=1))
    flat = optim.FlatProblem(f.chain, f.model, views, np.zeros(7, np.int32), np.arange(7, dtype=np.int32), f.intr, None,
                             f.view_pose, None)
    r0, J0 = helpers.oracle_eval(oracle, flat)
    with optim.ReprojHandle(flat) as h:
        h.eval()
        r1, J1 = h.eval_fetch()
        assert h.n_obs == sum(keep)
        nb = h.block_normal_eq()
        c1 = h.cost(1.0)
    assert np.abs(r0 - r1).max() <= 1e-9
    assert (np.abs(J0 - J1) / np.maximum(1.0, np.abs(J0))).max() <= 1e-9
    ref = helpers.oracle_block_normal_eq(oracle, flat)
    assert (np.abs(nb - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() <= 1e-10
    assert abs(c1 - helpers.oracle_cost(oracle, flat)) <= 1e-10 * max(1.0, c1)


@pytest.mark.parametrize("model", [0, 1])
@pytest.mark.parametrize("kind", ["intr", "ext", "bundle"])
def test_mode_b_block_normal_equations_and_cost(gpu_lib, oracle, kind, model):
    sc = SCENES[kind](model, noise_px=0.3)
    _perturb_intr(sc)
    ref = helpers.oracle_block_normal_eq(oracle, sc.flat)
    with optim.ReprojHandle(sc.flat) as h:
        nb = h.block_normal_eq()
        for delta in (1.0, -1.0, 25.0):
            c = h.cost(delta)
            c0 = helpers.oracle_cost(oracle, sc.flat, delta)
            assert abs(c - c0) <= 1e-11 * max(1.0, abs(c0))
    assert (np.abs(nb - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() <= 1e-10


def test_empty_view_and_bad_index_are_invalid_argument(gpu_lib):
    sc = synth.scene_intrinsics(5)
    f = sc.flat
    f.blk_offset[2] = f.blk_offset[1]  # block 1 empty -> "No observations provided"
    with pytest.raises(capi.CbaInvalidArgument):
        optim.ReprojHandle(f)
    sc = synth.scene_extrinsics(3, 2)
    sc.flat.blk_cam[0] = 5
    with pytest.raises(capi.CbaInvalidArgument):
        optim.ReprojHandle(sc.flat)


CASES = [
    ("intr", 0, {}, {}, 1e-9),
    ("intr", 0, dict(noise_px=0.2), {}, 1e-9),
    ("intr", 0, {}, dict(optimize_skew=1), 5e-9),  # free skew: conditioning ~1e7 puts rounding at the 1e-9 edge
    ("intr", 1, {}, {}, 1e-6),  # (narrow scene: the flat valley is demonstrated in test_scheimpflug_parity_gap_lies_in_the_flat_valley)
    ("ext", 0, {}, {}, 1e-9),
    ("ext", 0, dict(noise_px=0.2), dict(optimize_intrinsics=0), 1e-9),
    ("ext", 0, dict(noise_px=0.2), dict(optimize_extrinsics=0), 1e-9),
    ("ext", 1, {}, {}, 1e-9),
    ("ext", 1, dict(noise_px=0.2), {}, 1e-9),
    ("bundle", 0, {}, dict(optimize_intrinsics=1), 1e-9),
    ("bundle", 0, dict(noise_px=0.2), dict(optimize_intrinsics=0), 1e-9),
    ("bundle", 0, dict(noise_px=0.2), dict(optimize_intrinsics=1, huber_delta=-1.0), 1e-9),
    ("bundle", 0, dict(noise_px=0.2), dict(optimize_intrinsics=0, optimize_target_pose=0), 1e-9),
    ("bundle", 1, {}, dict(optimize_intrinsics=1), 1e-9),
]


@pytest.mark.parametrize("kind,model,skw,okw,tol", CASES)
def test_lm_solve_matches_oracle(gpu_lib, oracle, kind, model, skw, okw, tol):
    # 0.8 m x 0.56 m board (spacing 0.08) so conditioning does not eat the 1e-9 bar (see test_host_logic.py)
    mk = {"intr": lambda: synth.scene_intrinsics(12, model=model, spacing=0.08, **skw),
          "ext": lambda: synth.scene_extrinsics(6, 3, model=model, spacing=0.08, **skw),
          "bundle": lambda: synth.scene_bundle(16, 2, model=model, spacing=0.04, **skw)}[kind]
    for eps, bar in ((1e-12, tol), (1e-9, max(tol, 1e-7))):
        a, b = mk(), mk()
        o = options(epsilon=eps, **okw)
        sa = helpers.oracle_solve(oracle, a.flat, o)
        with optim.ReprojHandle(b.flat) as h:
            sb = h.solve(o)
        assert sb.termination == sa.termination
        assert abs(sb.iterations - sa.iterations) <= 2
        assert abs(sb.final_cost - sa.final_cost) <= 1e-9 * max(1.0, sa.final_cost) + 1e-15
        assert helpers.param_diff(a.flat, b.flat) <= bar, (eps, sb.report)


@pytest.mark.parametrize("seed", [3, 5])
def test_scheimpflug_well_conditioned_scene_meets_the_1e9_bar(gpu_lib, oracle, seed):
    """Scheimpflug at the north-star's parity bar (1e-9 relative): on a scene whose data determine every parameter
    (synth.scene_intrinsics_wide: 1.3 m board, tilts up to 45 degrees, depth spread, sensor tilt 0.2 rad) the HIP engine and the
    oracle agree to 1e-9.  Reference model: include/calib/models/scheimpflug.h:139-181."""
    a, b = synth.scene_intrinsics_wide(seed=seed), synth.scene_intrinsics_wide(seed=seed)
    o = options(epsilon=1e-12)
    sa = helpers.oracle_solve(oracle, a.flat, o, threads=16)
    sb = _gpu_solver(b.flat, o)
    assert sa.success and sb.success and abs(sb.iterations - sa.iterations) <= 1
    assert abs(sb.final_cost - sa.final_cost) <= 1e-12 * sa.final_cost
    assert helpers.param_diff(a.flat, b.flat) <= 1e-9, helpers.param_diff(a.flat, b.flat)


@pytest.mark.parametrize("seed,okw", [(3, dict(optimize_intrinsics=1)), (5, dict(optimize_intrinsics=1, huber_delta=-1.0)), (7, dict(optimize_intrinsics=1))])
def test_scheimpflug_bundle_chain_meets_the_1e9_bar_with_noise(gpu_lib, oracle, seed, okw):
    """The two-pose BUNDLE chain with the Scheimpflug model and 0.2 px noise at the north-star's bar: on a hand-eye bundle whose
    data determine the sensor tilt (synth.scene_bundle_wide: 0.65 m board at ~1 m, tilts up to 45 degrees, depth spread, tilt
    0.2 rad) the HIP engine and the oracle agree to 1e-9 - in fact to rounding.  (The EXTRINSIC chain meets it on the standard noisy
    scene: CASES.)  Reference: include/calib/models/scheimpflug.h:139-181 through src/estimation/residuals/bundleresidual.h:36-56."""
    a, b = synth.scene_bundle_wide(seed=seed), synth.scene_bundle_wide(seed=seed)
    o = options(epsilon=1e-12, **okw)
    sa = helpers.oracle_solve(oracle, a.flat, o, threads=16)
    sb = _gpu_solver(b.flat, o)
    assert sa.success and sb.success and abs(sb.iterations - sa.iterations) <= 1
    assert abs(sb.final_cost - sa.final_cost) <= 1e-12 * sa.final_cost
    assert helpers.param_diff(a.flat, b.flat) <= 1e-9, helpers.param_diff(a.flat, b.flat)


@pytest.mark.parametrize("seed", [7, 11, 13])
def test_scheimpflug_parity_gap_lies_in_the_flat_valley(gpu_lib, oracle, seed):
    """The reference's test geometry leaves the Scheimpflug tilt / principal point / focal length valley nearly flat (condition
    number of the Jacobi-scaled Hessian 1e8 .. 1e10).  Where the HIP engine and the oracle end further apart than rounding, the
    test DEMONSTRATES that this is conditioning and not arithmetic: costs equal to 1e-12, the difference has > 95 % of its scaled
    energy in the three weakest eigen-directions and a Rayleigh quotient within two orders of the smallest eigenvalue."""
    a, b = synth.scene_intrinsics(7, model=1, noise_px=0.2, seed=seed), synth.scene_intrinsics(7, model=1, noise_px=0.2, seed=seed)
    o = options(epsilon=1e-12, huber_delta=-1.0)
    sa = helpers.oracle_solve(oracle, a.flat, o)
    sb = _gpu_solver(b.flat, o)
    assert sa.success and sb.success and abs(sb.iterations - sa.iterations) <= 3
    assert abs(sb.final_cost - sa.final_cost) <= 1e-12 * sa.final_cost
    rep = helpers.weak_direction_report(oracle, a.flat, b.flat)
    gap = helpers.param_diff(a.flat, b.flat)
    assert rep["kappa"] > 1e8 and gap <= 1e-6, (rep, gap)
    if gap > 1e-10:
        assert rep["weak3_share"] > 0.95 and rep["rayleigh_over_lmin"] < 100.0, (rep, gap)


PINNED_FUZZ = [  # found by tools/fuzz_gpu.py in round 2 (profiles/r02_fuzz_*.json): parameter gaps far above the 1e-9 bar
    dict(kind="intr", model=1, seed=4976, noise=0.5, okw=dict(huber_delta=0.3, optimize_skew=0), nv=7, nc=2, grid=(7, 11)),
    dict(kind="intr", model=0, seed=705088, noise=0.1, okw=dict(huber_delta=3.0, optimize_skew=1), nv=6, nc=2, grid=(5, 5)),
    dict(kind="bundle", model=1, seed=32942, noise=0.1, okw=dict(huber_delta=-1.0, optimize_skew=0, optimize_intrinsics=1, optimize_extrinsics=1,
                                                                  optimize_target_pose=0), nv=8, nc=3, grid=(4, 9)),
    # round 3: a WELL-conditioned problem (only the view poses free, wrong constant intrinsics, every block in Huber's linear regime) on
    # which the iteration itself creeps: 200+ steps, stopped by the function tolerance a few 1e-6 short of each other
    dict(kind="ext", model=0, seed=1044805, noise=0.0, okw=dict(huber_delta=3.0, optimize_skew=1, optimize_intrinsics=0, optimize_extrinsics=0),
         nv=6, nc=2, grid=(5, 5)),
    # round 3, the one unexplained case of the 3000-case sweep before the stopping rule became rate-aware: only the target pose free
    # (6 unknowns, kappa 18), 18 vs 17 iterations, predicted cost gap 4.3 eps cost
    dict(kind="bundle", model=0, seed=210579, noise=0.1, okw=dict(huber_delta=0.3, optimize_skew=0, optimize_intrinsics=0, optimize_extrinsics=0,
                                                                   optimize_target_pose=1), nv=5, nc=1, grid=(9, 5)),
    # ... and of a second pair of 3000-case sweeps (seeds 21 / 22): a three-camera Scheimpflug bundle with free skew, kappa 2.8e9 -
    # one flat valley per camera, so 79 % of the gap sits in the three weakest directions and the rest in the next ones
    dict(kind="bundle", model=1, seed=69134, noise=0.1, okw=dict(huber_delta=0.3, optimize_skew=1, optimize_intrinsics=1, optimize_extrinsics=1,
                                                                  optimize_target_pose=0), nv=4, nc=3, grid=(6, 10)),
]


@pytest.mark.parametrize("rec", PINNED_FUZZ, ids=lambda r: f"{r['kind']}-{r['seed']}")
def test_pinned_fuzz_disagreements_are_conditioning_not_arithmetic(gpu_lib, oracle, hostmath, rec):
    """The random sweep's worst cases, pinned: the HIP engine and the oracle end 1e-6 .. 1e-3 apart in parameters.  The test does not
    accept that on faith: the gap must be BENIGN by the rule the sweep applies to every case (helpers.gap_is_benign) - same
    termination, no constant coordinate moved, a cost difference no larger than twice what the displacement predicts, and either
    >= 95 % of the Jacobi-scaled difference inside the three weakest eigen-directions of a Hessian with condition number > 1e6
    ("weak-direction") or >= 50 creeping iterations on both sides at costs equal to 1e-8 ("slow-convergence")."""
    rows, cols = rec["grid"]
    mk = {"intr": lambda: synth.scene_intrinsics(rec["nv"], rows=rows, cols=cols, spacing=0.08, model=rec["model"], noise_px=rec["noise"], seed=rec["seed"]),
          "ext": lambda: synth.scene_extrinsics(rec["nv"], max(2, rec["nc"]), rows=rows, cols=cols, spacing=0.08, model=rec["model"], noise_px=rec["noise"], seed=rec["seed"]),
          "bundle": lambda: synth.scene_bundle(rec["nv"] + 4, rec["nc"], rows=rows, cols=cols, spacing=0.04, model=rec["model"], noise_px=rec["noise"], seed=rec["seed"])}[rec["kind"]]
    a, b = mk(), mk()
    o = options(epsilon=1e-12, **rec["okw"])
    sa = helpers.oracle_solve(oracle, a.flat, o)
    with optim.ReprojHandle(b.flat) as h:
        sb = h.solve(o)
    assert sa.termination == sb.termination
    gap = helpers.param_diff(a.flat, b.flat)
    if gap > 1e-9:
        rep = helpers.solution_gap_report(oracle, hostmath, a.flat, b.flat, o)
        assert helpers.gap_is_benign(rep, sa.final_cost, sb.final_cost, (sa.iterations, sb.iterations), eps=1e-12), (gap, rep, sa.final_cost, sb.final_cost)


def test_bench_lm_geometry_against_the_oracle(gpu_lib, oracle, hostmath):
    """bench.py's LM workload (a 0.2 m board - 100 x 100 points at 0.002 m - seen from ~2 m) leaves the focal
    length weakly determined: the solve ends ~20 px from the generating intrinsics at 0.2 px noise.  That is the data, not the solver:
    on the same geometry at a size the oracle can hold (30 x 30 points, 24 views) the HIP engine and the oracle end at the same
    point - same termination and iteration count, costs equal to 1e-10, parameters to 1e-9 or a gap the classifier explains."""
    mk = lambda: synth.scene_intrinsics(24, rows=30, cols=30, spacing=0.2 / 30, seed=7, noise_px=0.2)
    a, b = mk(), mk()
    o = options(epsilon=1e-9)
    sa = helpers.oracle_solve(oracle, a.flat, o)
    with optim.ReprojHandle(b.flat) as h:
        sb = h.solve(o)
    assert sa.termination == sb.termination == capi.TERM_CONVERGENCE and abs(sa.iterations - sb.iterations) <= 1, (sa.report, sb.report)
    assert abs(sa.final_cost - sb.final_cost) <= 1e-10 * sa.final_cost
    gap = helpers.param_diff(a.flat, b.flat)
    if gap > 1e-9:
        rep = helpers.solution_gap_report(oracle, hostmath, a.flat, b.flat, o)
        assert helpers.gap_is_benign(rep, sa.final_cost, sb.final_cost, (sa.iterations, sb.iterations), eps=1e-9), (gap, rep)
    assert np.abs(b.flat.intr - b.gt_intr)[:, :2].max() > 1.0  # ... and yes: far from the generating focal lengths, on both sides


@pytest.mark.parametrize("kind,model,seed", [("intr", 0, 7), ("ext", 0, 9), ("intr", 0, 19), ("ext", 0, 23), ("intr", 0, 4)])
def test_projected_line_search_on_gpu_matches_the_oracle(gpu_lib, oracle, kind, model, seed):
    """Ceres' projected Armijo line search (bounds-constrained problems) through the HIP engine: from a rough start
    some steps fail the Armijo test at step size 1; the engine searches them (`Backend::line_eval`: k_scale_step, Mode R or Mode B +
    k_view_slope per sample) exactly where the oracle does.  With the resident kernel forced, the kernel hands the solve to the
    host-driven iteration (same result)."""
    a, b = helpers.rough_start_scene(kind, model, seed), helpers.rough_start_scene(kind, model, seed)
    o = options(epsilon=1e-10)
    sa = helpers.oracle_solve(oracle, a.flat, o)
    with optim.ReprojHandle(b.flat) as h:
        sb = h.solve(o)
        xs = h.solve_stats()
    assert xs["line_searches"] >= 1 and xs["line_search_evaluations"] >= xs["line_searches"], xs
    assert b"resident" not in bytes(sb.report)
    assert sb.termination == sa.termination and abs(sb.iterations - sa.iterations) <= 2, (sa.report, sb.report)
    assert abs(sb.final_cost - sa.final_cost) <= 1e-9 * sa.final_cost
    assert helpers.param_diff(a.flat, b.flat) <= 1e-7


def test_covariance_matches_oracle(gpu_lib, oracle):
    for mk, okw in ((lambda: synth.scene_intrinsics(6, noise_px=0.2), {}),
                    (lambda: synth.scene_extrinsics(4, 2, noise_px=0.2), {}),
                    (lambda: synth.scene_bundle(10, 2, noise_px=0.2), dict(optimize_intrinsics=1))):
        a, b = mk(), mk()
        o = options(**okw)
        helpers.oracle_solve(oracle, a.flat, o)
        cov0 = helpers.oracle_covariance(oracle, a.flat, o)
        with optim.ReprojHandle(b.flat) as h:
            h.solve(o)
            cov1 = h.covariance(o)
        assert cov0 is not None and cov1 is not None
        assert cov0.shape == cov1.shape
        d0 = np.abs(np.diag(cov0))
        assert np.array_equal(d0 == 0, np.diag(cov1) == 0)  # constant blocks: zero rows/cols on both sides
        nz = d0 > 0
        scale = np.sqrt(np.outer(d0[nz], d0[nz]))
        assert (np.abs(cov0 - cov1)[np.ix_(nz, nz)] / scale).max() <= 1e-5
        with optim.ReprojHandle(a.flat) as h:  # at the SAME point (the oracle's end point) the two assemblies differ by rounding only
            cov1a = h.covariance(o)
        assert (np.abs(cov0 - cov1a)[np.ix_(nz, nz)] / scale).max() <= 1e-7
        with optim.ReprojHandle(b.flat) as h:  # shared-block marginal = leading block of the full matrix, O(#views) work
            cs = h.covariance_shared(o)
        assert cs is not None and np.array_equal(cs, cov1[:cs.shape[0], :cs.shape[0]])
        if b.flat.n_views:  # per-view blocks on demand (cba_reproj_covariance_views) = the full matrix's diagonal blocks for those views
            V, ns = b.flat.n_views, cs.shape[0]
            sel = [V - 1, 0, V // 2]
            with optim.ReprojHandle(b.flat) as h:
                cv = h.covariance_views(o, sel)
            for k, v in enumerate(sel):
                rows = list(range(ns + 4 * v, ns + 4 * v + 4)) + list(range(ns + 4 * V + 3 * v, ns + 4 * V + 3 * v + 3))
                ref = cov1[np.ix_(rows, rows)]
                assert np.abs(cv[k] - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300)


def test_large_random_scene_properties(gpu_lib):
    """Size-independent properties at a size the oracle would take minutes for: J^T r from Mode A
    equals the Mode B gradient; cost equals 1/2 |r|^2 without loss; evaluation is deterministic."""
    sc = synth.scene_intrinsics(40, rows=50, cols=50, noise_px=0.2)
    with optim.ReprojHandle(sc.flat) as h:
        h.eval()
        r, J = h.eval_fetch()
        nb = h.block_normal_eq()
        nb2 = h.block_normal_eq()
        c = h.cost(-1.0)
    p = J.shape[1]
    nh = p * (p + 1) // 2
    g_modeb = nb[:, nh:nh + p]
    for b in range(sc.flat.n_blocks):
        lo, hi = 2 * sc.flat.blk_offset[b], 2 * sc.flat.blk_offset[b + 1]
        g = J[lo:hi].T @ r[lo:hi]
        assert np.abs(g - g_modeb[b]).max() <= 1e-9 * max(1.0, np.abs(g).max())
    assert abs(c - 0.5 * float(r @ r)) <= 1e-10 * c
    assert np.array_equal(nb, nb2)


# ---- the reference's own ground-truth-recovery tests, through the product's mirror API -------------
from tests import test_oracle_kat as kat  # noqa: E402


def _gpu_solver(flat, o):
    with optim.ReprojHandle(flat) as h:
        return h.solve(o)


def _gpu_cov(flat, o):
    with optim.ReprojHandle(flat) as h:
        return h.covariance(o)


@pytest.mark.parametrize("name", [k for k, v in kat.KAT.items() if v["kind"] == "intrinsics"])
def test_reference_kat_intrinsics_on_gpu(gpu_lib, name):
    kat.solve_kat_intrinsics(kat.KAT[name], _gpu_solver)


@pytest.mark.parametrize("name", [k for k, v in kat.KAT.items() if v["kind"] == "bundle"])
def test_reference_kat_bundle_on_gpu(gpu_lib, name):
    kat.solve_kat_bundle(kat.KAT[name], _gpu_solver)


@pytest.mark.parametrize("name", [k for k, v in kat.KAT.items() if v["kind"] == "extrinsics"])
def test_reference_kat_extrinsics_on_gpu(gpu_lib, name):
    kat.solve_kat_extrinsics(kat.KAT[name], _gpu_solver, _gpu_cov)


def test_mirror_api_end_to_end(gpu_lib):
    """optimize_intrinsics / optimize_extrinsics / optimize_bundle with the reference's signatures."""
    from calibration_amd.geometry import pose_to_matrix
    from tests.planar_seed import estimate_planar_pose

    sc = kat.KAT["intrinsics_noskew"]
    views = [np.asarray(v) for v in sc["views"]]
    cam0 = np.asarray(sc["cam_init"])
    res = optim.optimize_intrinsics(views, cam0, [estimate_planar_pose(v, cam0[:5]) for v in views],
                                    optim.IntrinsicsOptimOptions(num_radial=3, optimize_skew=False))
    assert res.core.success and np.abs(res.camera[:4] - np.asarray(sc["cam_gt"])[:4]).max() <= 1e-6
    assert res.core.final_cost < 1e-6 and len(res.c_se3_t) == len(views) and res.view_errors == []
    # (the 100-px-wide board of this reference scene makes J^T J numerically rank deficient: like the
    # reference, the result then simply carries no covariance)
    assert res.core.covariance is None or res.core.covariance.shape == (10 + 7 * len(views),) * 2
    sc = kat.KAT["bundle_two_cameras"]
    obs = [optim.BundleObservation(np.asarray(o["view"]), np.asarray(o["b_T_g"]), o["cam"]) for o in sc["obs"]]
    res = optim.optimize_bundle(obs, [np.asarray(c) for c in sc["cams_init"]], [np.asarray(T) for T in sc["g_T_c_init"]],
                                np.asarray(sc["b_T_t_init"]), optim.BundleOptions(optimize_intrinsics=False, optimize_target_pose=False))
    for X, Tg in zip(res.g_se3_c, sc["g_T_c_gt"]):
        assert np.linalg.norm(X[:3, 3] - np.asarray(Tg)[:3, 3]) < 1e-3
    sc = kat.KAT["extrinsics_poses"]
    res = optim.optimize_extrinsics([[np.asarray(v) for v in mv] for mv in sc["views"]], [np.asarray(c) for c in sc["cams_init"]],
                                    [np.asarray(T) for T in sc["c_T_r_init"]], [np.asarray(T) for T in sc["r_T_t_init"]],
                                    optim.ExtrinsicOptions(optimize_intrinsics=False))
    assert res.core.final_cost < 1e-6 and len(res.r_se3_t) == 3


# ---- AX = XB on the GPU ------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["axxb_refine", "stdrng/axxb_refine"])
def test_reference_kat_axxb_on_gpu(gpu_lib, name):
    from calibration_amd.geometry import rotation_angle

    sc = kat.KAT[name]
    X_gt, X0 = np.asarray(sc["X_gt"]), np.asarray(sc["X_init"])
    res = optim.optimize_handeye([np.asarray(T) for T in sc["b_T_g"]], [np.asarray(T) for T in sc["c_T_t"]], X0,
                                 optim.OptimOptions(optimizer=3, max_iterations=60, huber_delta=1.0))
    e0r = np.rad2deg(rotation_angle(X0[:3, :3].T @ X_gt[:3, :3]))
    e1r = np.rad2deg(rotation_angle(res.g_se3_c[:3, :3].T @ X_gt[:3, :3]))
    e1t = np.linalg.norm(res.g_se3_c[:3, 3] - X_gt[:3, 3])
    assert e1r < e0r and e1r < sc["tol_rot_deg"] and e1t < sc["tol_trans"]
    assert res.core.covariance is not None and res.core.covariance.shape == (7, 7)


@pytest.mark.parametrize("n_poses,noise", [(14, 0.0), (60, 0.3), (301, 0.3)])
def test_handeye_gpu_matches_oracle(gpu_lib, oracle, n_poses, noise):
    from calibration_amd.geometry import pose_from_matrix

    bTg, cTt, X_gt, X0 = helpers.handeye_scene(n_poses, seed=11, noise_rot_deg=noise, noise_trans=0.002 if noise else 0.0)
    pairs = np.ascontiguousarray(helpers.build_all_pairs(bTg, cTt, 0.5))
    o = options(epsilon=1e-12)
    xa = pose_from_matrix(X0)
    sa, ca = capi.CbaSummary(), np.zeros((7, 7))
    assert oracle.orc_axxb_solve(len(pairs), capi.dptr(pairs), capi.dptr(xa), C.byref(o), C.byref(sa), capi.dptr(ca)) == 0
    pb = np.stack([pose_from_matrix(T) for T in bTg])
    pc = np.stack([pose_from_matrix(T) for T in cTt])
    xb = pose_from_matrix(X0)
    sb, cb = capi.CbaSummary(), np.zeros((7, 7))
    capi.check(gpu_lib, gpu_lib.cba_optimize_handeye(n_poses, capi.dptr(pb), capi.dptr(pc), capi.dptr(xb), C.byref(o), C.byref(sb), capi.dptr(cb)))
    assert sb.termination == sa.termination and abs(sb.iterations - sa.iterations) <= 1
    assert f"{len(pairs)} pairs".encode() in sb.report
    assert np.abs(xa - xb).max() <= 1e-9
    assert abs(sa.final_cost - sb.final_cost) <= 1e-9 * max(1.0, sa.final_cost) + 1e-18
    assert np.abs(ca - cb).max() <= 1e-6 * np.abs(ca).max()


def test_handeye_rccl_transport_one_rank_equals_the_plain_entry_point(gpu_lib):
    """cba_estimate_and_optimize_handeye_rccl (BASELINE configs[3] names 8 GPUs): the 29 sums of every evaluation go through
    ncclAllReduce in device memory.  A one-GPU box can only run a communicator of one rank - the code path (communicator set-up,
    in-place collective on the evaluation's stream, one copy back, tear-down) is the multi-rank one; the sharding arithmetic is
    covered over gloo with 2 - 4 ranks (tests/test_multirank_gloo.py).  handeye.cpp:60-87."""
    from calibration_amd.geometry import pose_from_matrix

    bTg, cTt, _X_gt, _X0 = helpers.handeye_scene(40, seed=11, noise_rot_deg=0.05, noise_trans=0.0005)
    ref = optim.estimate_and_optimize_handeye(bTg, cTt, 1.0, optim.OptimOptions(epsilon=1e-12))
    res = optim.optimize_handeye_rccl(bTg, cTt, optim.rccl_unique_id(), 1, 0, init_gripper_se3_ref=None, min_angle_deg=1.0,
                                      options=optim.OptimOptions(epsilon=1e-12), device=0)
    assert res.core.success == ref.core.success and res.core.final_cost == ref.core.final_cost
    assert np.array_equal(pose_from_matrix(res.g_se3_c), pose_from_matrix(ref.g_se3_c))
    assert np.array_equal(res.core.covariance, ref.core.covariance)


def test_handeye_degenerate_is_runtime_error_on_gpu(gpu_lib):
    with pytest.raises(capi.CbaError) as ei:
        optim.optimize_handeye([np.eye(4)] * 5, [np.eye(4)] * 5, np.eye(4))
    assert ei.value.status == capi.CBA_ERR_RUNTIME and "No valid motion pairs" in ei.value.message


@pytest.mark.parametrize("blocked,variant", [(0, 0), (0, 3), (1, 0), (1, 5)])
def test_mode_a_layout_and_tuning_variants(gpu_lib, oracle, monkeypatch, blocked, variant):
    """Every k_eval build (whole-array columns / tile-blocked output, nt stores, tiles per wave) gives
    the same numbers through cba_reproj_eval_fetch."""
    monkeypatch.setenv("CBA_EVAL_BLOCKED", str(blocked))
    monkeypatch.setenv("CBA_EVAL_VARIANT", str(variant))
    sc = synth.scene_extrinsics(5, 3, rows=13, cols=21, noise_px=0.3)
    r0, J0 = helpers.oracle_eval(oracle, sc.flat)
    with optim.ReprojHandle(sc.flat) as h:
        h.eval()
        r1, J1 = h.eval_fetch()
    assert np.abs(r0 - r1).max() <= 1e-9
    assert (np.abs(J0 - J1) / np.maximum(1.0, np.abs(J0))).max() <= 1e-9


# ---- fp32 per-observation arithmetic (BASELINE config 5: fp32 kernel vs fp64 tolerance study) -------
@pytest.mark.parametrize("model", [0, 1])
@pytest.mark.parametrize("kind", ["intr", "ext", "bundle"])
def test_fp32_mode_a_error_vs_fp64_oracle(gpu_lib, oracle, kind, model):
    """fp32 residual/Jacobian against the fp64 oracle.  Pixels are ~1e3 with an fp32 ulp of 6e-5 px there,
    so the residual floor is ~1e-4 px (SURVEY.md §7 "fp32 study"); Jacobian entries hold ~1e-5 relative."""
    sc = SCENES[kind](model, noise_px=0.3)
    _perturb_intr(sc)
    r0, J0 = helpers.oracle_eval(oracle, sc.flat)
    with optim.ReprojHandle(sc.flat) as h:
        h.set_scalar(1)
        h.eval()
        r1, J1 = h.eval_fetch_f32()
        with pytest.raises(capi.CbaError):
            h.eval_fetch()
    assert r1.dtype == np.float32
    assert np.abs(r0 - r1).max() <= 3e-4
    assert (np.abs(J0 - J1) / np.maximum(1.0, np.abs(J0))).max() <= 2e-4
    assert np.sqrt(np.mean(((J0 - J1) / np.maximum(1.0, np.abs(J0))) ** 2)) <= 5e-6


@pytest.mark.parametrize("model", [0, 1])
def test_fp32_lm_deviation_from_fp64(gpu_lib, oracle, model):
    """LM with fp32 per-observation arithmetic (fp64 accumulation) vs the fp64 oracle on noisy data: the
    minimiser moves by far less than its own statistical uncertainty (0.2 px noise); asserted bounds:
    focal lengths / principal point within 2e-2 px, cost within 1e-5 relative."""
    mk = lambda: synth.scene_intrinsics(40, rows=20, cols=20, spacing=0.04, model=model, noise_px=0.2)
    a, b = mk(), mk()
    o = options()
    sa = helpers.oracle_solve(oracle, a.flat, o)
    with optim.ReprojHandle(b.flat) as h:
        h.set_scalar(1)
        sb = h.solve(o)
    assert sb.success and sa.success
    assert abs(sb.final_cost - sa.final_cost) <= 1e-4 * sa.final_cost  # the fp32 cost itself carries ~1e-4 px residual noise
    d = np.abs(a.flat.intr - b.flat.intr).reshape(-1)
    if model == 1:
        # Finding of the study: on this small problem (16 000 observations) tau, fx and the principal point trade
        # off along a nearly flat valley (condition ~1e8); fp32's 1e-7 relative noise moves the minimiser along it
        # by up to ~3.5 px in fx and ~0.08 rad in tau while the cost agrees to 1e-5.  Only the cost is asserted;
        # at the C5 size (1e7 observations) the same deviation is ~1e-3 px.
        print("fp32-vs-fp64 LM parameter deviation (Scheimpflug):", d)
        return
    # measured (pinhole): fx 1.6e-3, fy 1.7e-3, cx 1.6e-2, cy 5e-3 px; k1 2e-5, k2 6e-4, k3 6e-3 (k3 is barely
    # observable on a 0.8 m board), p1 8e-7, p2 3e-6
    assert d[:4].max() <= 5e-2, d
    assert d[5] <= 1e-3 and d[6] <= 1e-2 and d[7] <= 5e-2 and d[8:10].max() <= 1e-4, d
    print("fp32-vs-fp64 LM parameter deviation:", d)


# ---- planar pose by variable projection on the GPU (a5) ----------------------------------------------
def test_reference_kat_planar_pose_on_gpu(gpu_lib):
    """planarpose_test.cpp:96-146 and :148-211 through the mirror API."""
    view, true, init = helpers.planar_pose_scene()
    res = optim.optimize_planar_pose(view, helpers.PLANAR_K, init, optim.PlanarPoseOptions(num_radial=0))
    assert res.reprojection_error < 1e-3
    assert np.linalg.norm(res.pose[:3, :3] - true[:3, :3]) <= 1e-1 * np.linalg.norm(true[:3, :3])
    sc = true[2, 3] / res.pose[2, 3]
    assert np.linalg.norm(true[:3, 3] - sc * res.pose[:3, 3]) < 0.1
    view, true, init = helpers.planar_pose_scene(distort=True)
    res = optim.optimize_planar_pose(view, helpers.PLANAR_K, init, optim.PlanarPoseOptions(num_radial=1))
    assert res.reprojection_error < 1e-2
    assert np.linalg.norm(res.pose[:3, :3] - true[:3, :3]) <= 1e-1 * np.linalg.norm(true[:3, :3])
    assert len(res.distortion) == 3 and abs(res.distortion[0] - 0.1) <= 0.2


def test_planar_pose_batch_matches_oracle(gpu_lib, oracle):
    from calibration_amd.geometry import pose_from_matrix

    rng = np.random.default_rng(4)
    cam = synth.camera_gt(0)
    K = np.ascontiguousarray(cam[:5])
    views, inits = [], []
    for i in range(37):
        T = synth.random_view_poses(1, rng, dist=1.5)[0]
        grid = synth.make_target_grid(5 + i % 4, 6 + i % 3, 0.05)
        views.append(synth.render_view(cam, T, grid, noise_px=0.2, rng=rng))
        inits.append(synth.perturb_pose(T, rng, 1.5, 0.02))
    for nr in (0, 2):
        # epsilon 1e-12 on both sides: the bar measures arithmetic parity, not where each solver happened to stop (module docstring)
        out = optim.optimize_planar_pose_batch(views, K, inits, optim.PlanarPoseOptions(core=optim.OptimOptions(epsilon=1e-12), num_radial=nr))
        o = options(epsilon=1e-12)
        for i, (vw, T0) in enumerate(zip(views, inits)):
            X, Y, u, v = (np.ascontiguousarray(vw[:, k]) for k in range(4))
            p, s, d, rms, cov = helpers.pose6_of(T0), capi.CbaSummary(), np.zeros(nr + 2), C.c_double(), np.zeros((6, 6))
            assert oracle.orc_planar_pose_solve(len(vw), capi.dptr(X), capi.dptr(Y), capi.dptr(u), capi.dptr(v), capi.dptr(K), nr, capi.dptr(p),
                                                C.byref(o), C.byref(s), capi.dptr(d), C.byref(rms), capi.dptr(cov)) == 0
            r = out[i]
            assert r.core.success == bool(s.success) and abs(r.core.iterations - s.iterations) <= 1
            assert abs(r.reprojection_error - rms.value) <= 1e-9
            # the north-star's bar (planarpose.cpp:39-57) ... or the demonstration that the gap is not arithmetic: the two end points
            # are closer than the solvers' own stopping rule resolves (one of them took one step more: the quadratic model prices the
            # whole displacement below 4 eps cost), or it lies in the weakest directions of an ill-conditioned Hessian
            pg = helpers.pose6_of(r.pose)
            gap = np.abs(pg - p).max()
            if gap > 1e-9:
                rep = helpers.planar_pose_gap_report(oracle, vw, K, nr, p, pg)
                ca, cb = len(vw) * rms.value ** 2, len(vw) * r.reprojection_error ** 2  # 1/2 sum r^2
                assert gap <= 1e-5 and helpers.gap_category(rep, ca, cb, eps=1e-12) in ("stopping-resolution", "weak-direction"), (i, nr, gap, rep)
            else:
                assert np.abs(r.distortion - d).max() <= 1e-8
            # (the covariance is (J^T J)^-1 ssr / dof AT the end point: where the two end points differ by the stopping resolution, so does it)
            assert r.core.covariance is not None and np.abs(r.core.covariance - cov).max() <= (1e-6 if gap <= 1e-9 else 1e-4) * np.abs(cov).max()


def test_shared_and_distinct_target_point_lists(gpu_lib, oracle):
    """X, Y deduplication: views with identical object_xy lists share one device copy, views with different
    lists (dropped corners, reordered points, another target) do not; results are the same either way."""
    sc = synth.scene_intrinsics(9, rows=9, cols=14, spacing=0.05, noise_px=0.3)
    f = sc.flat
    views = []
    for b in range(f.n_blocks):
        lo, hi = f.blk_offset[b], f.blk_offset[b + 1]
        vw = np.stack([f.X[lo:hi], f.Y[lo:hi], f.u[lo:hi], f.v[lo:hi]], axis=1)
        if b % 3 == 1:
            vw = vw[::-1].copy()  # same points, different order
        if b % 3 == 2:
            vw = np.delete(vw, [5, 17, 40], axis=0)  # dropped corners
        views.append(vw)
    flat = optim.FlatProblem(f.chain, f.model, views, np.zeros(9, np.int32), np.arange(9, dtype=np.int32), f.intr, None, f.view_pose, None)
    r0, J0 = helpers.oracle_eval(oracle, flat)
    ref = helpers.oracle_block_normal_eq(oracle, flat)
    with optim.ReprojHandle(flat) as h:
        h.eval()
        r1, J1 = h.eval_fetch()
        nb = h.block_normal_eq()
        c = h.cost(1.0)
    assert np.abs(r0 - r1).max() <= 1e-9
    assert (np.abs(J0 - J1) / np.maximum(1.0, np.abs(J0))).max() <= 1e-9
    assert (np.abs(nb - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() <= 1e-10
    assert abs(c - helpers.oracle_cost(oracle, flat)) <= 1e-10 * c


def test_single_rank_transports_reproduce_the_plain_solve(gpu_lib):
    """The two all-reduce transports of the multi-GPU path, with one rank (all a 1-GPU box allows): the RCCL
    communicator inside libcalibba (ncclCommInitRank + ncclAllReduce on the engine's stream) and the host
    callback must leave the solve bit-identical to the transport-free one (a sum over one rank is the identity)."""
    o = options(epsilon=1e-10)

    def run(setup):
        sc = synth.scene_extrinsics(6, 3, noise_px=0.2, spacing=0.08)
        with optim.ReprojHandle(sc.flat) as h:
            setup(h)
            s = h.solve(o)
        return sc.flat, s

    f0, s0 = run(lambda h: h.set_lm_mode(0))  # the all-reduce lives in the host-driven iteration: compare like with like
    f1, s1 = run(lambda h: h.init_rccl(optim.rccl_unique_id(), 1, 0))
    calls = []

    def cb(arr):
        calls.append(arr.size)

    f2, s2 = run(lambda h: h.set_allreduce(cb, 1, 0))
    assert s0.success and s1.success and s2.success
    for f, s in ((f1, s1), (f2, s2)):
        assert s.iterations == s0.iterations and s.final_cost == s0.final_cost
        assert np.array_equal(f.intr, f0.intr) and np.array_equal(f.cam_pose, f0.cam_pose)
        assert np.array_equal(f.view_pose, f0.view_pose)


# ---- optimize_homography: one wavefront per view, whole LM in-kernel ------------------------------------------
def test_reference_kat_homography_on_gpu(gpu_lib):
    """homography_test.cpp:50-145 through the mirror API."""
    H = np.eye(3)
    H[0, 2], H[1, 2] = 10.0, -5.0
    xy = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], float)
    view = np.c_[xy, helpers.apply_homography(H, xy)]
    r = optim.optimize_homography(view, helpers.dlt_homography(view))
    assert r.core.success and helpers.is_approx(r.homography, H, 1e-6)
    view, H = helpers.homography_scene(50, 0.1)
    r = optim.optimize_homography(view, helpers.dlt_homography(view))
    assert r.core.success and helpers.is_approx(r.homography, H, 1e-2)
    assert r.core.covariance is not None and r.core.covariance.shape == (8, 8)
    view, H = helpers.homography_scene(100, 0.0, n_outliers=30)
    r = optim.optimize_homography(view, helpers.dlt_homography(view[:100]))
    assert r.core.success and helpers.is_approx(r.homography, H, 1e-2)
    with pytest.raises(ValueError, match="At least 4"):
        optim.optimize_homography(view[:3], np.eye(3))


def test_homography_batch_matches_oracle(gpu_lib, oracle):
    """A ragged batch (4 .. 3000 correspondences, with/without noise, outliers and loss) in one launch vs the oracle."""
    rng = np.random.default_rng(11)
    sizes = [4, 5, 63, 64, 65, 127, 128, 129, 500, 3000] + [int(k) for k in rng.integers(6, 400, 30)]
    for delta in (1.0, -1.0):
        views, inits = [], []
        for i, n in enumerate(sizes):
            noise = 0.0 if i % 3 == 0 else 0.3
            nout = (n // 10) if (i % 4 == 1 and delta > 0) else 0
            view, _ = helpers.homography_scene(n, noise, n_outliers=nout, seed=100 + i)
            H0 = helpers.dlt_homography(view[:n]) * (1 + 1e-3)
            H0[2, 2] = 1.0
            views.append(view)
            inits.append(H0)
        res = optim.optimize_homography_batch(views, inits, optim.OptimOptions(huber_delta=delta))
        o = options(huber_delta=delta)
        for view, H0, r in zip(views, inits, res):
            X, Y, u, v = (np.ascontiguousarray(view[:, k]) for k in range(4))
            h, s, cov = H0.reshape(9).copy(), capi.CbaSummary(), np.zeros((8, 8))
            assert oracle.orc_homography_solve(len(view), capi.dptr(X), capi.dptr(Y), capi.dptr(u), capi.dptr(v), capi.dptr(h), C.byref(o),
                                               C.byref(s), capi.dptr(cov)) == 0
            assert bool(s.success) == r.core.success and abs(s.iterations - r.core.iterations) <= 1
            assert np.abs(h.reshape(3, 3) - r.homography).max() <= 1e-9 * max(1.0, np.abs(h).max()), len(view)
            assert abs(s.final_cost - r.core.final_cost) <= 1e-9 * max(1.0, s.final_cost)
            if np.any(cov) and r.core.covariance is not None and s.final_cost > 1e-12:
                assert np.abs(cov - r.core.covariance).max() <= 1e-6 * np.abs(cov).max()


# ---- optimize_intrinsics_semidlt: per-view wave kernels + host arrow/Woodbury LM ---------------------------------------------
@pytest.mark.parametrize("case", [dict(nr=2, noise=0.0), dict(nr=2, noise=0.2), dict(nr=3, noise=0.2, okw=dict(optimize_skew=1)),
                                  dict(nr=2, noise=0.2, bounds=True), dict(nr=2, noise=0.2, fixed=[(1, 0.0)]), dict(nr=0, noise=0.2),
                                  dict(nr=1, noise=0.2, okw=dict(huber_delta=-1.0))])
def test_semidlt_gpu_matches_oracle(gpu_lib, oracle, case):
    nr = case["nr"]
    d, kgt, agt = helpers.semidlt_scene(5, noise=case["noise"], nr=nr)
    o = options(epsilon=1e-12, **case.get("okw", {}))
    lo, hi = (None, None)
    if case.get("bounds"):
        lo, hi = [kgt[0] - 200, kgt[1] - 200, kgt[2] - 30, kgt[3] - 30, -0.01], [kgt[0] + 200, kgt[1] - 25.0, kgt[2] + 30, kgt[3] + 30, 0.01]
    gpu_lib.cba_optimize_intrinsics_semidlt.argtypes = helpers.SEMIDLT_SOLVE_ARGS
    sta, ka, pa, sa, da, va, ca = helpers.semidlt_solve(oracle.orc_semidlt_solve, d, nr, o, lo, hi, case.get("fixed"))
    stb, kb, pb, sb, db, vb, cb = helpers.semidlt_solve(gpu_lib.cba_optimize_intrinsics_semidlt, d, nr, o, lo, hi, case.get("fixed"))
    assert sta == 0 and stb == 0
    assert sa.termination == sb.termination == capi.TERM_CONVERGENCE and abs(sa.iterations - sb.iterations) <= 2
    assert abs(sa.final_cost - sb.final_cost) <= 1e-9 * max(1.0, sa.final_cost)
    assert np.abs(ka - kb).max() <= 1e-9 * np.abs(ka).max() and np.abs(pa - pb).max() <= 1e-9   # the north-star's bar
    assert np.abs(da - db).max() <= 1e-9 * max(1.0, np.abs(da).max()) and np.abs(va - vb).max() <= 1e-9
    if case["noise"] > 0:
        dg = np.sqrt(np.abs(np.diag(ca)))
        nz = dg > 0
        assert np.any(nz) and (np.abs(ca - cb)[np.ix_(nz, nz)] / np.outer(dg[nz], dg[nz])).max() <= 1e-5
    else:
        assert np.abs(kb[:4] - kgt[:4]).max() <= 1e-6 and np.abs(db - agt).max() <= 1e-7 and vb.max() <= 1e-8


def test_semidlt_mirror_api_and_large_problem(gpu_lib):
    """Through the mirror of the reference signature: ground-truth recovery on 60 views x 900 points (a size the Jet oracle cannot
    hold), the < 4 views contract (default result, no exception: intrinsicssemidlt.cpp:163-166) and active bounds."""
    from calibration_amd.geometry import pose_to_matrix

    d, kgt, agt = helpers.semidlt_scene(60, rows=30, cols=30, noise=0.0, nr=2, seed=11)
    views = [np.c_[d["X"][a:b], d["Y"][a:b], d["u"][a:b], d["v"][a:b]] for a, b in zip(d["off"][:-1], d["off"][1:])]
    seeds = [pose_to_matrix(p) for p in d["poses0"]]
    opt = optim.IntrinsicsOptimOptions(core=optim.OptimOptions(epsilon=1e-12, compute_covariance=False), num_radial=2)
    r = optim.optimize_intrinsics_semidlt(views, d["kappa0"], seeds, opt)
    assert r.core.success, r.core.report
    assert np.abs(r.camera[:4] - kgt[:4]).max() <= 1e-6 and np.abs(r.distortion - agt).max() <= 1e-7
    assert max(r.view_errors) <= 1e-8 and len(r.c_se3_t) == 60
    for T, pg in zip(r.c_se3_t, d["poses_gt"]):
        assert np.abs(T - pose_to_matrix(pg)).max() <= 1e-7
    few = optim.optimize_intrinsics_semidlt(views[:3], d["kappa0"], seeds[:3], opt)
    assert not few.core.success and few.c_se3_t == []
    b = optim.CalibrationBounds(fx_max=kgt[0] - 10.0, fy_max=2000.0, cx_max=1280.0, cy_max=720.0)
    rb = optim.optimize_intrinsics_semidlt(views, d["kappa0"], seeds, opt, bounds=b)
    assert rb.camera[0] == kgt[0] - 10.0


def test_semidlt_rccl_transport_one_rank_equals_the_plain_entry_point(gpu_lib):
    """cba_optimize_intrinsics_semidlt_rccl: the sharded evaluator (split pass-1 sum / alpha kernels, row-gather of the per-view
    table, the sums reduced by ncclAllReduce in device memory) on a communicator of ONE rank - all a one-GPU box can run - against
    the single-GPU entry point, whose pass-1 sum it reproduces in the same order: equal to the last bit.  The multi-rank
    arithmetic is covered over gloo (tests/test_multirank_gloo.py).  intrinsicssemidlt.cpp:155-191."""
    from calibration_amd.geometry import pose_from_matrix, pose_to_matrix

    d, _kgt, _agt = helpers.semidlt_scene(8, rows=7, cols=9, noise=0.2, nr=2, seed=5)
    views = [np.c_[d["X"][a:b], d["Y"][a:b], d["u"][a:b], d["v"][a:b]] for a, b in zip(d["off"][:-1], d["off"][1:])]
    seeds = [pose_to_matrix(p) for p in d["poses0"]]
    opt = optim.IntrinsicsOptimOptions(core=optim.OptimOptions(epsilon=1e-12, compute_covariance=True), num_radial=2)
    ref = optim.optimize_intrinsics_semidlt(views, d["kappa0"], seeds, opt)
    res = optim.optimize_intrinsics_semidlt_sharded(views, 0, 8, d["kappa0"], seeds, 1, 0, rccl_id=optim.rccl_unique_id(), opts=opt, device=0)
    assert res.core.success == ref.core.success and res.core.iterations == ref.core.iterations and res.core.final_cost == ref.core.final_cost
    assert np.array_equal(res.camera, ref.camera) and np.array_equal(res.distortion, ref.distortion)
    assert np.array_equal(np.stack([pose_from_matrix(T) for T in res.c_se3_t]), np.stack([pose_from_matrix(T) for T in ref.c_se3_t]))
    assert np.array_equal(res.core.covariance, ref.core.covariance) and res.view_errors == ref.view_errors


# ---- batched estimate_planar_pose on the GPU -------------------------------------------------------------
def test_planar_seed_batch_on_gpu(gpu_lib):
    from tests.planar_seed import estimate_planar_pose
    from tests.test_host_logic import _seed_views

    for noise in (0.0, 0.3):
        cam, views, poses = _seed_views(n_views=40, noise=noise)
        views.append(views[0][:3])  # < 4 points -> identity
        got = optim.estimate_planar_pose_batch(views, cam[:5])
        assert np.array_equal(got[-1], np.eye(4))
        for view, T, Tgt in zip(views[:-1], got, poses):
            Tr = estimate_planar_pose(view, cam[:5])
            assert np.abs(T - Tr).max() <= (1e-9 if len(view) > 4 else 1e-7)
            if noise == 0.0:
                assert np.abs(T - Tgt).max() <= 1e-8
    # the reference's calling pattern: optimize_intrinsics_semidlt seeds every view itself (intrinsicssemidlt.cpp:37-40)
    d, kgt, agt = helpers.semidlt_scene(8, noise=0.0, nr=2, seed=21)
    vs = [np.c_[d["X"][a:b], d["Y"][a:b], d["u"][a:b], d["v"][a:b]] for a, b in zip(d["off"][:-1], d["off"][1:])]
    r = optim.optimize_intrinsics_semidlt(vs, d["kappa0"], None, optim.IntrinsicsOptimOptions(core=optim.OptimOptions(epsilon=1e-12)))
    assert r.core.success and np.abs(r.camera[:4] - kgt[:4]).max() <= 1e-6 and np.abs(r.distortion - agt).max() <= 1e-7
    with pytest.raises(ValueError, match="out of range"):
        optim.optimize_intrinsics_semidlt(vs, d["kappa0"], None, fixed_distortion_indices=[7])


# ---- Tsai-Lenz seed + estimate_and_optimize_handeye on the GPU -------------------------------------------------------------------
def test_tsai_lenz_seed_and_combined_entry_point_on_gpu(gpu_lib):
    from calibration_amd.geometry import rotation_angle

    for n, noise in ((20, 0.0), (20, 0.05), (150, 0.05)):
        bTg, cTt, X, _ = helpers.handeye_scene(n, seed=123, noise_rot_deg=noise, noise_trans=noise * 1e-3)
        T = optim.estimate_handeye_dlt(bTg, cTt, 1.0)
        assert np.abs(T - helpers.tsai_lenz_dlt(bTg, cTt, 1.0)).max() <= 1e-10
        assert np.rad2deg(rotation_angle(T[:3, :3].T @ X[:3, :3])) < 10 and np.linalg.norm(T[:3, 3] - X[:3, 3]) < 0.03  # see test_host_logic
    with pytest.raises(RuntimeError, match="No valid motion pairs"):
        optim.estimate_handeye_dlt([np.eye(4)] * 5, [np.eye(4)] * 5, 2.0)
    # handeye_test.cpp:101-152 through estimate_and_optimize_handeye: 18 frames, Huber 1.0, rot < 0.05 deg, trans < 2 mm
    bTg, cTt, X, _ = helpers.handeye_scene(18, seed=2024, noise_rot_deg=0.02, noise_trans=2e-4)
    r = optim.estimate_and_optimize_handeye(bTg, cTt, 1.0, optim.OptimOptions(max_iterations=60, huber_delta=1.0))
    assert r.core.success
    assert np.rad2deg(rotation_angle(r.g_se3_c[:3, :3].T @ X[:3, :3])) < 0.05 and np.linalg.norm(r.g_se3_c[:3, 3] - X[:3, 3]) < 2e-3


def test_full_size_c2_properties(gpu_lib):
    """BASELINE configs[1] at FULL size (1000 views x 10 000 points = 1e7 observations; the oracle would need minutes): the
    size-independent properties of the path.  Mode A's J^T r equals Mode B's gradient block by block, Mode B's |r|^2 and the
    cost kernel agree with 1/2 |r|^2 from Mode A, evaluation is deterministic, and the LM recovers the camera to the
    statistical accuracy of 0.2 px noise on 1e7 observations."""
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
    with optim.ReprojHandle(sc.flat) as h:
        h.eval()
        r, J = h.eval_fetch()
        nb = h.block_normal_eq()
        assert np.array_equal(nb, h.block_normal_eq())
        c = h.cost(-1.0)
        p = J.shape[1]
        nh = p * (p + 1) // 2
        off = sc.flat.blk_offset
        worst_g, worst_h = 0.0, 0.0
        for b in range(0, sc.flat.n_blocks, 37):  # every 37th block: 28 blocks x 20 000 rows
            lo, hi = 2 * off[b], 2 * off[b + 1]
            Jb, rb = J[lo:hi], r[lo:hi]
            g = Jb.T @ rb
            worst_g = max(worst_g, np.abs(g - nb[b, nh:nh + p]).max() / max(1.0, np.abs(g).max()))
            H = Jb.T @ Jb
            worst_h = max(worst_h, np.abs(H[np.triu_indices(p)] - nb[b, :nh]).max() / np.abs(H).max())
        assert worst_g <= 1e-9 and worst_h <= 1e-10
        assert abs(nb[:, -1].sum() - float(r @ r)) <= 1e-10 * float(r @ r)
        assert abs(c - 0.5 * float(r @ r)) <= 1e-10 * c
        del r, J
        o = options(compute_covariance=0)
        s = h.solve(o)
        assert s.success and s.iterations <= 15
        cs = h.covariance_shared(o)
    err = np.abs(sc.flat.intr - sc.gt_intr).reshape(-1)
    sig = np.sqrt(np.diag(cs))
    act = sig > 0
    assert (err[act] <= 6.0 * sig[act]).all(), (err, sig)  # within 6 sigma of the engine's own covariance
    assert err[:4].max() < 0.2  # and well under a fifth of a pixel in fx, fy, cx, cy


def test_full_size_c4_handeye_chain(gpu_lib):
    """BASELINE configs[3] at FULL size, one GPU: 2000 robot poses -> 1 998 961 motion pairs formed on the device, Tsai-Lenz seed
    (estimate_handeye_dlt), AX = XB refinement, then the joint bundle over 2000 poses x 4 cameras x 88 points.  The oracle needs
    minutes at this size: the checks are the reference's own ground-truth tolerances (handeye_test.cpp:148-151: 0.05 deg, 2 mm;
    bundle_test.cpp:66-80: 1e-3 rad / m at its noise-free data, here with 0.2 px noise 6 sigma of the engine's covariance),
    Mode A's J^T r / J^T J equal to Mode B's on sampled blocks, and bitwise repeatability."""
    from calibration_amd.geometry import rotation_angle

    bTg, cTt, X_gt, _ = helpers.handeye_scene(2000, seed=2024, noise_rot_deg=0.1, noise_trans=0.001)
    oo = optim.OptimOptions(max_iterations=60, huber_delta=1.0)
    r1 = optim.estimate_and_optimize_handeye(bTg, cTt, 1.0, oo)
    r2 = optim.estimate_and_optimize_handeye(bTg, cTt, 1.0, oo)
    assert r1.core.success and np.array_equal(r1.g_se3_c, r2.g_se3_c) and r1.core.final_cost == r2.core.final_cost
    assert np.rad2deg(rotation_angle(r1.g_se3_c[:3, :3].T @ X_gt[:3, :3])) < 0.05
    assert np.linalg.norm(r1.g_se3_c[:3, 3] - X_gt[:3, 3]) < 2e-3
    # The linear seed alone: the reference solves the Tsai-Lenz system skew(alpha + beta) x = beta - alpha with alpha, beta the
    # LOG vectors of the motions and takes exp(x) (handeyedlt.cpp:84-99), which returns about half of X's rotation angle (10 deg
    # in this scene) whatever the pose count or the noise - the restatement (helpers.tsai_lenz_dlt) gives 4.987 deg at 18 ... 300
    # poses.  The device seed reproduces that, and the refinement above is what recovers X.
    seed = optim.estimate_handeye_dlt(bTg, cTt, 1.0)
    assert 4.9 < np.rad2deg(rotation_angle(seed[:3, :3].T @ X_gt[:3, :3])) < 5.1 and np.linalg.norm(seed[:3, 3] - X_gt[:3, 3]) < 0.05

    sc = synth.scene_bundle(2000, 4, noise_px=0.2, seed=2024)
    assert sc.flat.n_blocks == 8000 and sc.flat.n_obs == 704000
    with optim.ReprojHandle(sc.flat) as h:
        h.eval()
        r, J = h.eval_fetch()
        nb = h.block_normal_eq()
        assert np.array_equal(nb, h.block_normal_eq())
        p = J.shape[1]
        nh = p * (p + 1) // 2
        off = sc.flat.blk_offset
        for b in range(0, sc.flat.n_blocks, 97):
            lo, hi = 2 * off[b], 2 * off[b + 1]
            Jb, rb = J[lo:hi], r[lo:hi]
            g, H = Jb.T @ rb, Jb.T @ Jb
            assert np.abs(g - nb[b, nh:nh + p]).max() <= 1e-9 * max(1.0, np.abs(g).max())
            assert np.abs(H[np.triu_indices(p)] - nb[b, :nh]).max() <= 1e-10 * np.abs(H).max()
        assert abs(h.cost(-1.0) - 0.5 * float(r @ r)) <= 1e-10 * float(r @ r)
        start = (sc.flat.intr.copy(), sc.flat.cam_pose.copy(), sc.flat.target_pose.copy())
        o = options(optimize_intrinsics=1)
        s = h.solve(o)
        assert s.success and s.iterations <= 20, s.report
        cov = h.covariance(o)
        first = (sc.flat.intr.copy(), sc.flat.cam_pose.copy(), sc.flat.target_pose.copy())
        h.set_params(intr=start[0], cam_pose=start[1], target_pose=start[2])
        s2 = h.solve(o)
        assert s2.iterations == s.iterations and s2.final_cost == s.final_cost
        assert all(np.array_equal(a, b) for a, b in zip(first, (sc.flat.intr, sc.flat.cam_pose, sc.flat.target_pose)))
    # the four cameras within 6 sigma of the engine's covariance (intrinsics block of each camera) and far inside a pixel
    assert cov is not None
    err = np.abs(sc.flat.intr - sc.gt_intr)
    assert err[:, :4].max() < 0.5, err
    for c in range(4):  # hand-eye poses: camera 0 is the gauge in the reference's problem when ... all are free here
        q, qg = sc.flat.cam_pose[c], sc.gt_cam_pose[c]
        from calibration_amd.geometry import pose_to_matrix
        T, Tg = pose_to_matrix(q), pose_to_matrix(qg)
        assert rotation_angle(T[:3, :3].T @ Tg[:3, :3]) < 1e-3 and np.linalg.norm(T[:3, 3] - Tg[:3, 3]) < 1e-3, (c, T, Tg)


def test_full_size_c3_properties(gpu_lib, lm_mode):
    """BASELINE configs[2] at FULL size — 4000 views x 8 cameras x 5000 points = 1.6e8 observations, 32 000 residual blocks,
    optimize_extrinsics (src/estimation/optim/extrinsics.cpp:174-196) — through the size-independent properties of the path
    (the oracle would need an hour).  Mode A's 59 GB output stays in HBM: sampled blocks come back through the block-range
    fetch and must reproduce Mode B's (moment-form) J^T J and J^T r; evaluation and the whole solve are bitwise repeatable; the
    LM converges at the reference's epsilon = 1e-9 and lands within 6 sigma of the engine's own shared covariance."""
    if lm_mode == "resident":
        pytest.skip("one pass at this size (the resident kernel never takes a problem of 1.6e8 observations)")
    sc = synth.scene_extrinsics(4000, 8, rows=50, cols=100, spacing=0.008, noise_px=0.2, seed=137)
    assert sc.flat.n_obs == 160_000_000 and sc.flat.n_blocks == 32_000
    start = (sc.flat.intr.copy(), sc.flat.cam_pose.copy(), sc.flat.view_pose.copy())
    with optim.ReprojHandle(sc.flat) as h:
        h.eval()
        nb = h.block_normal_eq()
        assert np.array_equal(nb, h.block_normal_eq())  # no atomics anywhere: bitwise repeatable
        p = h.local_columns
        nh = p * (p + 1) // 2
        worst_g = worst_h = 0.0
        for b in (0, 1, 7, 4097, 15_999, 16_000, 23_456, 31_999):
            r, J = h.eval_fetch_blocks(b, b + 1)
            assert r.shape == (10_000,) and J.shape == (10_000, p)
            g, H = J.T @ r, J.T @ J
            worst_g = max(worst_g, np.abs(g - nb[b, nh:nh + p]).max() / max(1.0, np.abs(g).max()))
            d = np.sqrt(np.diag(H))
            ok = np.outer(d, d)[np.triu_indices(p)] > 0
            worst_h = max(worst_h, (np.abs(H[np.triu_indices(p)] - nb[b, :nh])[ok] / np.outer(d, d)[np.triu_indices(p)][ok]).max())
            assert abs(float(r @ r) - nb[b, -1]) <= 1e-11 * nb[b, -1]
        assert worst_g <= 1e-9 and worst_h <= 1e-10, (worst_g, worst_h)
        c = h.cost(-1.0)
        assert abs(c - 0.5 * nb[:, -1].sum()) <= 1e-10 * c
        o = options(compute_covariance=0)
        s = h.solve(o)
        assert s.success and s.iterations <= 15, s.report
        first = (sc.flat.intr.copy(), sc.flat.cam_pose.copy(), sc.flat.view_pose.copy())
        xs = h.solve_stats()
        # one exchange point per LM step (a no-op on a single rank): the initial system, one per trial point, one more only for a
        # rejected step, a radius miss, or a step accepted after a plain trial
        if os.environ.get("CBA_LM_SPECULATE", "1") != "0":  # the plain protocol (a selectable knob) exchanges twice per step
            assert xs["speculative_steps"] >= 1 and xs["speculation_hits"] >= 1
            assert xs["allreduce_calls"] == 1 + s.iterations + xs["speculation_misses"] + xs["rejected_steps"] + xs[
                "line_search_evaluations"] + (s.successful_steps - xs["speculation_hits"] - xs["speculation_misses"])
        cs = h.covariance_shared(o)
        assert cs is not None, gpu_lib.cba_last_error().decode()
        # per-view pose covariance on demand at this size (the reference-layout matrix would be 28 000^2): view 0 is the gauge
        # (constant: zeros), the others symmetric positive semi-definite with a millimetre-scale translation sigma or better
        cv = h.covariance_views(o, [0, 1999, 3999])
        assert cv is not None and not cv[0].any()
        for blk in cv[1:]:
            assert np.abs(blk - blk.T).max() <= 1e-12 * np.abs(blk).max() and np.linalg.eigvalsh(blk).min() >= -1e-12 * np.abs(blk).max()
            assert 0.0 < np.sqrt(np.diag(blk)[4:]).max() < 1e-3
        # the same solve again from the same start on the same handle: bitwise identical end state
        h.set_params(intr=start[0], cam_pose=start[1], view_pose=start[2])
        s2 = h.solve(o)
        assert (s2.iterations, s2.final_cost) == (s.iterations, s.final_cost)
        assert all(np.array_equal(a, b) for a, b in zip(first, (sc.flat.intr, sc.flat.cam_pose, sc.flat.view_pose)))
    err = np.abs(sc.flat.intr - sc.gt_intr)
    assert err[:, :4].max() < 0.1  # fx, fy, cx, cy: 2e7 observations per camera at 0.2 px
    # every intrinsic within 6 sigma of the engine's own shared-block covariance.  cba_reproj_covariance_shared tests the reciprocal
    # condition number of the reduced system (1e-14, Ceres' own threshold) instead of a row-count-dependent QR tolerance, so it
    # answers at this size.  The matrix is (J^T W J)^-1 as the reference's four main stages return it (unscaled, ceresutils.h:117-123
    # scales only planar pose / homography / semi-DLT); with 0.2 px noise and the per-BLOCK Huber weight w ~ delta / |r_block| ~ 0.05
    # of a 10 000-residual block its sigma over-states the estimator's standard deviation (0.2 px sqrt(w) ~ 0.045 of it): the
    # check is conservative by that factor, and still far tighter than the absolute bounds below
    sig = np.sqrt(np.diag(cs))[:err.size]
    act = sig > 0
    assert act.sum() >= 8 * 9 and (err.reshape(-1)[act] <= 6.0 * sig[act]).all(), (err.reshape(-1)[act] / sig[act]).max()
    assert err[:, 5].max() < 2e-3 and err[:, 8:10].max() < 1e-4  # (absolute: k1 is strongly correlated with k2, k3 on this field of view)


def test_exact_c1_shape_against_the_oracle(gpu_lib, oracle):
    """BASELINE configs[0] exactly: one pinhole intrinsic refinement, 20 views x 88 points (8 x 11 target, 0.02 m pitch, the
    reference's own test geometry, intrinsics_optimize_test.cpp:8-61), with the reference's defaults (epsilon = 1e-9) and at
    epsilon = 1e-12: same termination and iteration count as the oracle, parameters to 1e-9."""
    for eps, tol in ((1e-9, 1e-7), (1e-12, 1e-9)):
        a = synth.scene_intrinsics(20, noise_px=0.2)
        b = synth.scene_intrinsics(20, noise_px=0.2)
        assert a.flat.n_obs == 20 * 88
        o = options(epsilon=eps)
        sa = helpers.oracle_solve(oracle, a.flat, o, threads=16)
        with optim.ReprojHandle(b.flat) as h:
            h.eval()
            r1, J1 = h.eval_fetch()
            sb = h.solve(o)
        r0, J0 = helpers.oracle_eval(oracle, synth.scene_intrinsics(20, noise_px=0.2).flat)
        assert np.abs(r0 - r1).max() <= 1e-9 and (np.abs(J0 - J1) / np.maximum(1.0, np.abs(J0))).max() <= 1e-9
        assert sb.termination == sa.termination and abs(sb.iterations - sa.iterations) <= 1, (sa.report, sb.report)
        assert abs(sb.final_cost - sa.final_cost) <= 1e-10 * sa.final_cost
        assert helpers.param_diff(a.flat, b.flat) <= tol, helpers.param_diff(a.flat, b.flat)


def test_full_size_c5_fp32_study(gpu_lib, lm_mode):
    """BASELINE configs[4] at FULL size (Scheimpflug intrinsics, 1000 views x 10 000 points = 1e7 observations, 0.2 px noise):
    the fp32 kernels against the fp64 ones.  fp32 Mode A rows agree with fp64 to a few fp32 ulps of a pixel coordinate, the
    fp32 LM ends at the same cost to 1e-5 relative and moves the intrinsics far less than their own standard deviation —
    while staying orders of magnitude above the 1e-9 parity bar, which therefore remains an fp64-only claim."""
    if lm_mode == "resident":
        pytest.skip("one pass at this size")
    res = {}
    for scalar in (0, 1):
        sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2, model=capi.CAMERA_SCHEIMPFLUG, seed=5)
        with optim.ReprojHandle(sc.flat) as h:
            h.set_scalar(scalar)
            h.eval()
            if scalar:
                r, J = h.eval_fetch_f32()
            else:
                r, J = h.eval_fetch_blocks(0, 8)
            s = h.solve(options(compute_covariance=0))
            cs = h.covariance_shared(options()) if not scalar else None
        res[scalar] = (sc, np.asarray(r[:2 * 80_000], dtype=np.float64), np.asarray(J[:2 * 80_000], dtype=np.float64), s, cs)
    (s64, r64, J64, st64, cs), (s32, r32, J32, st32, _) = res[0], res[1]
    assert st64.success and st32.success
    assert np.abs(r32 - r64).max() <= 2e-4  # fp32 ulp at ~1e3 px is 6e-5 px
    assert (np.abs(J32 - J64) / np.maximum(1.0, np.abs(J64))).max() <= 1e-4
    assert abs(st32.final_cost - st64.final_cost) <= 1e-5 * st64.final_cost
    d = np.abs(s32.flat.intr - s64.flat.intr).reshape(-1)
    sig = np.sqrt(np.diag(cs))[:d.size]
    act = sig > 0
    assert (d[act] <= 1.0 * sig[act]).all(), (d[act] / sig[act]).max()  # fp32 moves every intrinsic by less than one sigma ...
    assert helpers.rel_diff(s64.flat.intr, s32.flat.intr) > 1e-8               # ... and by far more than the fp64 parity bar


def test_c3_shaped_moment_mode_b_equals_direct_mode_b(gpu_lib, monkeypatch):
    """BASELINE configs[2] shape at 1/20 of the views (200 views x 8 cameras x 5000 points = 8e6 observations): the moment
    form of Mode B (k_normal_eq_mom + k_mom_expand) against the direct 22-column form, block by block, and the LM end result."""
    def run(moments):
        monkeypatch.setenv("CBA_MODEB_MOMENTS", str(moments))
        sc = synth.scene_extrinsics(200, 8, rows=50, cols=100, spacing=0.008, noise_px=0.2, seed=137)
        with optim.ReprojHandle(sc.flat) as h:
            nb = h.block_normal_eq()
            s = h.solve(options(compute_covariance=0))
        return sc, nb, s

    sa, na, ra = run(1)
    sb, nb_, rb = run(0)
    p = helpers.local_cols(sa.flat)
    nh = p * (p + 1) // 2
    diag = np.sqrt(na[:, [i * p - i * (i - 1) // 2 for i in range(p)]])  # sqrt(H_ii) per block
    iu = np.triu_indices(p)
    scale = diag[:, iu[0]] * diag[:, iu[1]]
    ok = scale > 0
    assert (np.abs(na[:, :nh] - nb_[:, :nh])[ok] / scale[ok]).max() <= 1e-10
    assert np.abs(na[:, nh:] - nb_[:, nh:]).max() <= 1e-9 * np.abs(nb_[:, nh:]).max()
    assert ra.success and rb.success and ra.iterations == rb.iterations
    assert abs(ra.final_cost - rb.final_cost) <= 1e-10 * rb.final_cost
    assert helpers.param_diff(sa.flat, sb.flat) <= 1e-9
    assert np.abs(sa.flat.intr - sa.gt_intr)[:, :4].max() < 0.5


@pytest.mark.parametrize("chain", ["intr", "ext"])
def test_mode_b_tile_boundaries(gpu_lib, oracle, chain):
    """Mode B / Mode R tile edges (a tile is 64 lanes x 16 passes = 1024 observations; the loads of pass k+1 are prefetched):
    blocks of 63, 64, 65, 1023, 1024, 1025, 2049 and 3000 points, for the direct form (INTRINSIC chain) and the moment
    form (EXTRINSIC chain), against the oracle's Jets."""
    sizes = [63, 64, 65, 1023, 1024, 1025, 2049, 3000]
    if chain == "intr":
        sc = synth.scene_intrinsics(len(sizes), rows=55, cols=55, spacing=0.012, noise_px=0.3)
        f = sc.flat
        views = [np.stack([f.X[f.blk_offset[b]:f.blk_offset[b] + n], f.Y[f.blk_offset[b]:f.blk_offset[b] + n],
                           f.u[f.blk_offset[b]:f.blk_offset[b] + n], f.v[f.blk_offset[b]:f.blk_offset[b] + n]], axis=1)
                 for b, n in enumerate(sizes)]
        flat = optim.FlatProblem(f.chain, f.model, views, np.zeros(len(sizes), np.int32), np.arange(len(sizes), dtype=np.int32),
                                 f.intr, None, f.view_pose, None)
    else:
        sc = synth.scene_extrinsics(4, 2, rows=55, cols=55, spacing=0.012, noise_px=0.3)
        f = sc.flat
        views = [np.stack([f.X[f.blk_offset[b]:f.blk_offset[b] + n], f.Y[f.blk_offset[b]:f.blk_offset[b] + n],
                           f.u[f.blk_offset[b]:f.blk_offset[b] + n], f.v[f.blk_offset[b]:f.blk_offset[b] + n]], axis=1)
                 for b, n in enumerate(sizes)]
        flat = optim.FlatProblem(f.chain, f.model, views, f.blk_cam, f.blk_view, f.intr, f.cam_pose, f.view_pose, None)
    flat.intr[...] = flat.intr * (1 + 0.01 * np.random.default_rng(5).uniform(-1, 1, flat.intr.shape))
    with optim.ReprojHandle(flat) as h:
        nb = h.block_normal_eq()
        c1 = h.cost(1.0)
    ref = helpers.oracle_block_normal_eq(oracle, flat)
    assert (np.abs(nb - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() <= 1e-10
    assert abs(c1 - helpers.oracle_cost(oracle, flat)) <= 1e-10 * max(1.0, c1)


@pytest.mark.parametrize("kind,n_cams,n", [("ext", 9, 144), ("ext", 10, 160), ("bundle", 8, 134), ("ext", 8, 128)])
def test_wide_reduced_systems_through_the_controller(gpu_lib, oracle, kind, n_cams, n, lm_mode):
    """The controller kernel's memory forms (lm_ctl.hip): up to 128 shared columns the reduced matrix and every short per-column
    array live in LDS (8-camera rig: 128, the LDS limit itself); 129 .. 136 the matrix moves to global memory while the short arrays
    stay in LDS (8-camera hand-eye bundle: 6 + 8 x 16 = 134); above, everything is in global memory (9 and 10 cameras: 144, 160).
    Each against the oracle at the 1e-9 bar and against the host-side form of the iteration (same decisions)."""
    if lm_mode == "resident":
        pytest.skip("wider than the resident kernel takes")
    mk = (lambda: synth.scene_extrinsics(5, n_cams, spacing=0.08, noise_px=0.2, seed=5)) if kind == "ext" else \
         (lambda: synth.scene_bundle(14, n_cams, spacing=0.04, noise_px=0.2, seed=5))
    a, b, c = mk(), mk(), mk()
    o = options(epsilon=1e-12, optimize_intrinsics=1) if kind == "bundle" else options(epsilon=1e-12)
    sa = helpers.oracle_solve(oracle, a.flat, o)
    with optim.ReprojHandle(b.flat) as h:
        sb = h.solve(o)
        xb = h.solve_stats()
    with optim.ReprojHandle(c.flat) as h:
        h.set_lm_mode(3)
        sc = h.solve(o)
        xc = h.solve_stats()
    assert helpers.local_cols(b.flat) >= 22
    assert sb.termination == sa.termination and abs(sb.iterations - sa.iterations) <= 1, (sa.report, sb.report)
    assert abs(sb.final_cost - sa.final_cost) <= 1e-10 * sa.final_cost
    assert helpers.param_diff(a.flat, b.flat) <= 1e-9
    assert (sc.termination, sc.iterations, sc.successful_steps) == (sb.termination, sb.iterations, sb.successful_steps), (sb.report, sc.report)
    assert {k: xc[k] for k in xc if k != "allreduce_doubles"} == {k: xb[k] for k in xb if k != "allreduce_doubles"}
    assert helpers.param_diff(b.flat, c.flat) <= 1e-11


from tests.test_host_logic import _sweep_cases  # noqa: E402


@pytest.mark.parametrize("idx,kind,seed,noise,okw", _sweep_cases(18, seed=77))
def test_random_option_sweep_on_gpu(gpu_lib, oracle, idx, kind, seed, noise, okw):
    """The stage-switch sweep of tests/test_host_logic.py through the HIP engine: same constant blocks, gauge rules and end point."""
    mk = {"intr": lambda: synth.scene_intrinsics(6, spacing=0.08, noise_px=noise, seed=seed),
          "ext": lambda: synth.scene_extrinsics(4, 3, spacing=0.08, noise_px=noise, seed=seed),
          "bundle": lambda: synth.scene_bundle(8, 2, spacing=0.04, noise_px=noise, seed=seed)}[kind]
    a, b = mk(), mk()
    o = options(epsilon=1e-12, **okw)
    sa = helpers.oracle_solve(oracle, a.flat, o)
    with optim.ReprojHandle(b.flat) as h:
        sb = h.solve(o)
    assert sb.termination == sa.termination, (sa.report, sb.report)
    assert abs(sb.iterations - sa.iterations) <= 2
    assert abs(sb.final_cost - sa.final_cost) <= 1e-8 * max(1.0, sa.final_cost) + 1e-14
    assert helpers.param_diff(a.flat, b.flat) <= (5e-8 if okw.get("optimize_skew") else 2e-9), okw


def test_non_finite_observations_fail_cleanly(gpu_lib):
    """A NaN pixel must end in termination FAILURE / success = false (as ceres::Solve would report), never in a hang or a
    crash — for the host-driven LM and for the in-kernel per-view solvers (whose loops are bounded by max_iterations)."""
    sc = synth.scene_extrinsics(4, 2, noise_px=0.1)
    sc.flat.u[5] = np.nan
    with optim.ReprojHandle(sc.flat) as h:
        s = h.solve(options())
    assert not s.success and s.termination == capi.TERM_FAILURE and s.iterations <= 5
    view, H = helpers.homography_scene(50, 0.1)
    bad = view.copy()
    bad[7, 2] = np.inf
    r = optim.optimize_homography_batch([view, bad], [helpers.dlt_homography(view)] * 2)
    assert r[0].core.success and not r[1].core.success
    pv, true, init = helpers.planar_pose_scene(distort=True, noise=0.1)
    badp = pv.copy()
    badp[3, 3] = np.nan
    rp = optim.optimize_planar_pose_batch([pv, badp], helpers.PLANAR_K, [init, init])
    assert rp[0].core.success and not rp[1].core.success
    bTg, cTt, X, X0 = helpers.handeye_scene(10)
    cTt[3] = cTt[3].copy()
    cTt[3][0, 3] = np.nan
    rh = optim.optimize_handeye(bTg, cTt, X0)
    assert not rh.core.success


def test_unobserved_camera_on_gpu(gpu_lib, oracle):
    """tests/test_host_logic.py::test_camera_without_observations_and_ragged_rig through the HIP engine: the solve leaves the idle
    camera untouched and agrees with the oracle; the covariance is reported rank deficient (empty), as ceres::Covariance would."""
    def make():
        sc = synth.scene_extrinsics(5, 3, spacing=0.08, noise_px=0.2)
        f = sc.flat
        keep = [b for b in range(f.n_blocks) if f.blk_cam[b] != 2 and not (f.blk_cam[b] == 1 and f.blk_view[b] in (0, 2))]
        views = [np.stack([f.X[f.blk_offset[b]:f.blk_offset[b + 1]], f.Y[f.blk_offset[b]:f.blk_offset[b + 1]],
                           f.u[f.blk_offset[b]:f.blk_offset[b + 1]], f.v[f.blk_offset[b]:f.blk_offset[b + 1]]], axis=1) for b in keep]
        return optim.FlatProblem(f.chain, f.model, views, f.blk_cam[keep], f.blk_view[keep], f.intr, f.cam_pose, f.view_pose, None)

    a, b = make(), make()
    idle = (b.intr[2].copy(), b.cam_pose[2].copy())
    o = options(epsilon=1e-12)
    sa = helpers.oracle_solve(oracle, a, o)
    with optim.ReprojHandle(b) as h:
        sb = h.solve(o)
        cov = h.covariance(o)
        o2 = options(epsilon=1e-12, optimize_intrinsics=0, optimize_extrinsics=0)
        cov2 = h.covariance(o2)
    assert sa.termination == sb.termination == capi.TERM_CONVERGENCE and helpers.param_diff(a, b) <= 2e-9
    assert np.array_equal(b.intr[2], idle[0]) and np.array_equal(b.cam_pose[2], idle[1])
    assert cov is None and cov2 is not None


def test_cost_reduction_above_4096_blocks(gpu_lib, oracle):
    """The cost kernel switches to a two-stage reduction above 4096 residual blocks (C3 has 32 000): 4400 blocks here, Huber
    active on part of them, against the oracle; and the LM still solves the problem."""
    sc = synth.scene_bundle(1100, 4, noise_px=0.4, seed=9)
    assert sc.flat.n_blocks > 4096
    with optim.ReprojHandle(sc.flat) as h:
        for delta in (1.0, -1.0, 8.0):
            c1 = h.cost(delta)
            assert abs(c1 - helpers.oracle_cost(oracle, sc.flat, delta)) <= 1e-10 * c1
        s = h.solve(options(compute_covariance=0, optimize_intrinsics=1))
    assert s.success


def test_end_to_end_seed_then_refine(gpu_lib):
    """The calling pattern of the reference's intrinsics facade (src/pipeline/facades/intrinsics.cpp:100-136): per-view pose
    seeds from a rough camera matrix (estimate_planar_pose), then the non-linear refinement, then the covariance — all on the
    device: batched DLT seeds -> semi-DLT (K + poses, distortion by variable projection) -> optimize_intrinsics with the full
    Brown-Conrady model.  planar_intrinsics_test.cpp:343-348 asks for +-5 px on such a pipeline; noise-free data must come back
    exactly."""
    from calibration_amd.geometry import pose_to_matrix

    for noise, tol in ((0.0, 1e-6), (0.3, 5.0)):
        sc = synth.scene_intrinsics(25, rows=9, cols=12, spacing=0.06, noise_px=noise, seed=31)
        f = sc.flat
        views = [np.c_[f.X[a:b], f.Y[a:b], f.u[a:b], f.v[a:b]] for a, b in zip(f.blk_offset[:-1], f.blk_offset[1:])]
        K0 = sc.gt_intr.reshape(-1)[:5] * np.array([0.95, 1.04, 1.01, 0.99, 1.0])
        seeds = optim.estimate_planar_pose_batch(views, K0)
        sd = optim.optimize_intrinsics_semidlt(views, K0, seeds, optim.IntrinsicsOptimOptions(core=optim.OptimOptions(compute_covariance=False), num_radial=3))
        assert sd.core.success
        r = optim.optimize_intrinsics(views, sd.camera, sd.c_se3_t)
        assert r.core.success and r.core.covariance is not None
        err = np.abs(r.camera - sc.gt_intr.reshape(-1))
        assert err[:4].max() <= tol and (noise > 0 or err[5:].max() <= 1e-7), (noise, err)
        if noise > 0:  # every parameter within 6 sigma of the engine's own covariance (k3 is weakly determined on this board)
            sig = np.sqrt(np.diag(r.core.covariance)[:10])
            ok = sig > 0
            assert (err[ok] <= 6 * sig[ok]).all(), (err, sig)


def test_homography_dlt_batch_then_refine_on_gpu(gpu_lib):
    """estimate_homography (DLT) -> optimize_homography, both batched on the device, as homography_test.cpp:50-93 chains them."""
    from tests.planar_seed import homography_dlt

    views, truth = [], []
    for i, n in enumerate([4, 5, 50, 64, 65, 400]):
        view, H = helpers.homography_scene(n, 0.0 if i % 2 == 0 else 0.1, seed=300 + i)
        views.append(view)
        truth.append(H)
    views.append(views[0][:3])  # < 4 correspondences: fit fails
    Hs, ok = optim.estimate_homography_batch(views)
    assert ok == [True] * 6 + [False] and np.array_equal(Hs[-1], np.eye(3))
    for view, H, Ht in zip(views[:-1], Hs, truth):
        Hr = homography_dlt(view[:, :2], view[:, 2:])
        # (as in the reference, only the NORMALISED solve has H22 = 1: T_dst^-1 Hn T_src is returned without a final rescale,
        #  and optimize_homography then reads its first 8 entries as they are — homographyestimator.cpp:70, 79-87; homography.cpp:79-84)
        assert np.abs(H - Hr).max() <= 1e-8 * np.abs(Hr).max()
    res = optim.optimize_homography_batch(views[:-1], Hs[:-1])
    for r, Ht, view in zip(res, truth, views[:-1]):
        assert r.core.success and helpers.is_approx(r.homography, Ht, 1e-2 if len(view) > 5 else 5e-2)


def test_eight_camera_rig_schur_contraction_on_mfma(gpu_lib, oracle, monkeypatch):
    """BASELINE config 3's rig (8 cameras: shared block 8 x 16 = 128 wide): the Schur contraction S -= sum_v Z_v^T Z_v runs on
    v_mfma_f64_16x16x4_f64 (k_schur_syrk_mfma) when the shared block is >= 64 wide.  LM parity against the dense oracle, and
    against the register-blocked VALU form (CBA_SYRK_MFMA=0)."""
    def run(mfma):
        monkeypatch.setenv("CBA_SYRK_MFMA", str(mfma))
        sc = synth.scene_extrinsics(11, 8, rows=6, cols=7, spacing=0.08, noise_px=0.2, seed=137)
        with optim.ReprojHandle(sc.flat) as h:
            s = h.solve(options(epsilon=1e-12, compute_covariance=0))
        return sc, s

    a = synth.scene_extrinsics(11, 8, rows=6, cols=7, spacing=0.08, noise_px=0.2, seed=137)
    sa = helpers.oracle_solve(oracle, a.flat, options(epsilon=1e-12, compute_covariance=0))
    (b, sb), (c, sc_) = run(1), run(0)
    for x, sx in ((b, sb), (c, sc_)):
        assert sx.termination == sa.termination and abs(sx.iterations - sa.iterations) <= 2
        assert abs(sx.final_cost - sa.final_cost) <= 1e-9 * sa.final_cost
        assert helpers.param_diff(a.flat, x.flat) <= 2e-9
    assert helpers.param_diff(b.flat, c.flat) <= 1e-10


def test_plain_c_example_runs(gpu_lib, tmp_path):
    """examples/c_api_demo.c: the C ABI used from plain C (no Python, no torch in the process)."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, libdir = str(tmp_path / "demo"), os.path.join(root, "calibration_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_api_demo.c"), "-L", libdir,
                    "-lcalibba", f"-Wl,-rpath,{libdir}", "-lm", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    vals = dict(zip(("fx", "fy", "cx", "cy"), [float(t) for t in r.stdout.split("fx")[1].replace("fy", " ").replace("cx", " ").replace("cy", " ").split("(")[0].split()]))
    assert abs(vals["fx"] - 1000) < 1e-5 and abs(vals["fy"] - 1005) < 1e-5 and abs(vals["cx"] - 640) < 1e-5 and abs(vals["cy"] - 360) < 1e-5
    assert "success 1" in r.stdout


@pytest.mark.parametrize("model", [0, 1])
@pytest.mark.parametrize("kind", ["intr", "ext", "bundle"])
def test_resident_kernel_agrees_with_host_driven_iteration(gpu_lib, kind, model, lm_mode):
    """The same solve through cba_reproj_set_lm_mode 0 and 2: identical decisions (termination, iteration and accepted-step
    counts), parameters equal to rounding, and the resident solve bitwise reproducible."""
    if lm_mode == "host":
        pytest.skip("compares both forms itself")
    sc = SCENES[kind](model, noise_px=0.3)
    _perturb_intr(sc)
    o = options(epsilon=1e-10, optimize_intrinsics=1)
    runs = {}
    for mode in (0, 2, 2):
        f = copy.deepcopy(sc.flat)
        with optim.ReprojHandle(f) as h:
            h.set_lm_mode(mode)
            s = h.solve(o)
            searched = h.solve_stats()["line_searches"]
            cov = h.covariance_shared(o)
        runs.setdefault(mode, []).append((s, f, cov, searched))
    (s0, f0, c0, ls0), = runs[0]
    (s1, f1, c1, ls1), (s2, f2, c2, _) = runs[2]
    assert b"resident" not in bytes(s0.report)
    # a bounds-constrained solve in which a step fails the Armijo test is handed back by the resident kernel to the host-driven
    # iteration (which runs Ceres' line search): then both runs are the host-driven form
    assert b"resident kernel" in bytes(s1.report) or (ls0 > 0 and ls1 == ls0), (s0.report, s1.report, ls0, ls1)
    assert (s1.termination, s1.iterations, s1.successful_steps) == (s0.termination, s0.iterations, s0.successful_steps)
    assert abs(s1.final_cost - s0.final_cost) <= 1e-12 * abs(s0.final_cost)
    assert abs(s1.initial_cost - s0.initial_cost) <= 1e-13 * abs(s0.initial_cost)
    tol = 1e-8 if model == 0 else 1e-7  # summation order only, amplified by the conditioning of k3 / tau
    for name in ("intr", "cam_pose", "view_pose", "target_pose"):
        a, b = getattr(f0, name), getattr(f1, name)
        if a is None:
            continue
        assert (np.abs(a - b) / np.maximum(1.0, np.abs(a))).max() <= tol, name
        assert np.array_equal(b, getattr(f2, name)), name
    assert s2.final_cost == s1.final_cost and s2.iterations == s1.iterations
    assert np.abs(c1 - c0).max() <= 1e-6 * np.abs(c0).max()


def test_resident_kernel_is_the_default_for_small_problems_only(gpu_lib, monkeypatch):
    monkeypatch.delenv("CBA_LM_RESIDENT", raising=False)
    small = synth.scene_intrinsics(7, noise_px=0.2)
    _perturb_intr(small)
    with optim.ReprojHandle(small.flat) as h:
        assert b"resident kernel" in bytes(h.solve(options()).report)
    big = synth.scene_intrinsics(7, rows=40, cols=40, noise_px=0.2)  # 11 200 observations: past the crossover
    _perturb_intr(big)
    with optim.ReprojHandle(big.flat) as h:
        assert b"resident" not in bytes(h.solve(options()).report)
        h.set_lm_mode(2)
        _perturb_intr(big)
        h.set_params()
        assert b"resident kernel" in bytes(h.solve(options()).report)


def test_block_cache_reuse_leaves_results_unchanged(gpu_lib):
    """Handles return their device / pinned blocks and their stream to a process-wide cache (block_cache.cpp).  A solve on
    recycled blocks (which hold another problem's stale data) must equal the solve on fresh ones bit for bit."""
    def run(n_views, seed):
        sc = synth.scene_intrinsics(n_views, noise_px=0.2, seed=seed)
        with optim.ReprojHandle(sc.flat) as h:
            s = h.solve(options())
            cov = h.covariance_shared(options())
        return sc.flat.intr.copy(), sc.flat.view_pose.copy(), s.final_cost, s.iterations, cov

    gpu_lib.cba_trim_cache()
    fresh = run(9, 3)
    run(14, 4)          # different sizes: other size classes enter the cache
    run(9, 5)           # same sizes, different data: the blocks `fresh` used now hold this problem's state
    again = run(9, 3)
    for a, b in zip(fresh, again):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    gpu_lib.cba_trim_cache()
    after_trim = run(9, 3)
    for a, b in zip(fresh, after_trim):
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize("kind", ["intr", "ext"])
def test_create_from_observation_records_equals_flat_arrays(gpu_lib, kind):
    """cba_reproj_create_aos reads {object_xy, image_uv} records in place (the memory of std::vector<PlanarObservation>):
    same device state as the flat-array creation — Mode A output, de-duplication and LM result bit for bit; ragged blocks,
    one block with its own target points."""
    sc = SCENES[kind](0, noise_px=0.3)
    f = sc.flat
    lo, hi = int(f.blk_offset[1]), int(f.blk_offset[2])
    f.X[lo:hi] += 1e-3  # block 1 no longer shares the target point list
    recs = [np.ascontiguousarray(np.stack([f.X[a:b], f.Y[a:b], f.u[a:b], f.v[a:b]], axis=1))
            for a, b in zip(f.blk_offset[:-1], f.blk_offset[1:])]
    _perturb_intr(sc)
    o = options(epsilon=1e-10, optimize_intrinsics=1)
    out = []
    for records in (None, recs):
        g = copy.deepcopy(f)
        with optim.ReprojHandle(g, records=records) as h:
            h.eval()
            r, J = h.eval_fetch()
            s = h.solve(o)
        out.append((r, J, g.intr.copy(), g.view_pose.copy(), s.final_cost, s.iterations))
    for a, b in zip(*out):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    with pytest.raises(ValueError):
        optim.ReprojHandle(copy.deepcopy(f), records=recs[:-1] + [np.zeros((0, 4))])  # record count differs from blk_offset


@pytest.mark.parametrize("kind", ["intr", "ext", "bundle"])
def test_hip_graph_replay_of_the_lm_stages_equals_plain_launches(gpu_lib, kind, monkeypatch, lm_mode):
    """The host-driven iteration switches a stage to a captured HIP graph after 200 plain uses (CBA_LM_GRAPH=1: at once).
    Capture + replay must reproduce the plain launches bit for bit, first solve (capture) and second solve (pure replay)."""
    if lm_mode == "resident":
        pytest.skip("graphs belong to the host-driven iteration")
    o = options(epsilon=1e-10, optimize_intrinsics=1)

    def run(graph):
        monkeypatch.setenv("CBA_LM_GRAPH", graph)
        sc = SCENES[kind](0, noise_px=0.3)
        _perturb_intr(sc)
        f = sc.flat
        init = [None if x is None else x.copy() for x in (f.intr, f.cam_pose, f.view_pose, f.target_pose)]
        out = []
        with optim.ReprojHandle(f) as h:
            for _ in range(2):
                h.set_params(*init)
                s = h.solve(o)
                out.append((s.iterations, s.final_cost, f.intr.copy(), None if f.view_pose is None else f.view_pose.copy()))
        return out

    plain, graph = run("0"), run("1")
    for a, b in zip(plain + plain[:1], graph + graph[1:]):
        assert a[0] == b[0] and a[1] == b[1]
        assert np.array_equal(a[2], b[2])
        assert (a[3] is None and b[3] is None) or np.array_equal(a[3], b[3])


def test_default_device_of_the_handle_less_entry_points(gpu_lib):
    """cba_set_device / cba_get_device: the device of the one-shot and batched entry points (one process per GPU sets its own)."""
    assert gpu_lib.cba_get_device() == 0
    capi.check(gpu_lib, gpu_lib.cba_set_device(0))
    n = gpu_lib.cba_device_count()
    assert gpu_lib.cba_set_device(n) == capi.CBA_ERR_INVALID_ARGUMENT and gpu_lib.cba_get_device() == 0
    assert gpu_lib.cba_set_device(-1) == capi.CBA_ERR_INVALID_ARGUMENT
