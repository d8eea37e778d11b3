"""CPU tier: the N > 1 path over gloo (world_size 2), no GPU.

Views are sharded across ranks (synth.shard_views), shared parameters replicated, and the host LM
driver issues its packed sum-all-reduces through the cba_allreduce_fn callback — the same driver,
packing and callback contract libcalibba uses on GPUs (where the per-rank arithmetic runs in HIP
kernels and the transport may be RCCL instead).  Here the per-rank arithmetic is the test-only CPU
backend.  Checks: every rank ends with identical shared parameters, they match the 1-rank solve to
1e-9 relative, and the gauge rule (global view 0 fixed) follows the shard that owns it.
"""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(kind):
    from tests import synth

    if kind == "intr":
        return synth.scene_intrinsics(11, spacing=0.08, noise_px=0.2)
    if kind == "ext":
        return synth.scene_extrinsics(7, 3, spacing=0.08, noise_px=0.2)
    if kind == "ext2":  # two views only: with three ranks one rank owns no view at all
        return synth.scene_extrinsics(2, 2, spacing=0.08, noise_px=0.2)
    return synth.scene_bundle(13, 2, spacing=0.04, noise_px=0.2)


def _worker(rank, world, port, kind, okw, outdir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from calibration_amd import capi
    from tests import synth
    from calibration_amd.capi import CbaSummary
    from tests import helpers

    hm = helpers.load_hostmath()
    flat = synth.shard_views(_scene(kind).flat, rank, world)
    calls = []

    def allreduce(buf, count, _user):
        arr = np.ctypeslib.as_array(buf, shape=(int(count),))
        t = torch.from_numpy(arr)
        dist.all_reduce(t)  # in-place sum on the shared buffer
        calls.append(int(count))
        return 0

    cb = capi.ALLREDUCE_FN(allreduce)
    d = flat.struct()
    s = CbaSummary()
    o = helpers.options(epsilon=1e-12, **okw)
    xs = (C.c_int64 * 8)()
    st = hm.hm_reproj_solve_ex(C.byref(d), C.byref(o), cb, None, world, rank, -1, C.byref(s), xs)
    assert st == 0, hm.hm_last_error()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), intr=flat.intr, cam=flat.cam_pose if flat.cam_pose is not None else np.zeros(0),
             view=flat.view_pose if flat.view_pose is not None else np.zeros(0),
             target=flat.target_pose if flat.target_pose is not None else np.zeros(0), first=flat.first_view_global,
             iters=s.iterations, cost=s.final_cost, term=s.termination, ncalls=len(calls), xs=np.array(list(xs)),
             accepted=s.successful_steps, sizes=np.array(calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,okw,world", [("intr", {}, 2), ("ext", {}, 2), ("ext", dict(optimize_intrinsics=0), 2),
                                             ("bundle", dict(optimize_intrinsics=1), 2), ("ext2", dict(optimize_intrinsics=0), 3)])
def test_two_ranks_match_one_rank(hostmath, tmp_path, kind, okw, world):
    import torch.multiprocessing as mp

    from calibration_amd import capi
    from calibration_amd.capi import CbaSummary
    from tests import helpers

    mp.spawn(_worker, args=(world, _free_port(), kind, okw, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    # single-rank reference with the same driver
    ref = _scene(kind).flat
    d = ref.struct()
    s = CbaSummary()
    o = helpers.options(epsilon=1e-12, **okw)
    assert hostmath.hm_reproj_solve(C.byref(d), C.byref(o), capi.ALLREDUCE_FN(), None, 1, 0, C.byref(s)) == 0
    for r in res:
        assert int(r["term"]) == s.termination
        assert abs(int(r["iters"]) - s.iterations) <= 1
        assert abs(float(r["cost"]) - s.final_cost) <= 1e-9 * max(1.0, s.final_cost)
        assert int(r["ncalls"]) >= 2
        assert helpers.rel_diff(ref.intr, r["intr"]) <= 1e-9
        if ref.cam_pose is not None:
            assert helpers.rel_diff(ref.cam_pose, r["cam"]) <= 1e-9
        if ref.target_pose is not None:
            assert helpers.rel_diff(ref.target_pose, r["target"]) <= 1e-9
        # ONE packed all-reduce per LM step.  Every trial point is one exchange (speculative steps carry
        # the next system with their statistics); beyond those: the initial system, one re-elimination per rejected step and per
        # accepted step whose radius was not the predicted one, one new system per step accepted after a plain (cost-only) trial.
        calls, _doubles, spec, hits, misses, rejected, _ls, ls_evals = (int(v) for v in r["xs"])
        iters, accepted = int(r["iters"]), int(r["accepted"])
        assert calls == int(r["ncalls"]) and spec >= 1 and hits >= 1 and hits + misses <= spec
        accepted_plain = accepted - hits - misses
        assert calls == 1 + iters + misses + rejected + accepted_plain + ls_evals
        # one collective per step, plus one per rejection / radius miss / line-search sample / step accepted after a plain trial (a step
        # evaluated the cheap way because it followed a rejection or was expected to end the solve: at most a few per solve)
        assert calls <= iters + 1 + misses + 2 * rejected + 2 * ls_evals + accepted_plain and accepted_plain <= 2 + rejected
    # replicated blocks are bit-identical across ranks (same all-reduced sums, same host arithmetic)
    for r in res[1:]:
        assert np.array_equal(res[0]["intr"], r["intr"]) and np.array_equal(res[0]["cam"], r["cam"])
    if ref.view_pose is not None:
        views = np.concatenate([r["view"].reshape(-1, 7) for r in res])
        assert helpers.rel_diff(ref.view_pose.reshape(-1, 7), views) <= 1e-9
        assert int(res[0]["first"]) == 0 and int(res[1]["first"]) == res[0]["view"].reshape(-1, 7).shape[0]


# ---- semi-DLT with the views sharded: the exchange protocol of semidlt.hip's evaluator on the host build, 2 processes over gloo ---
def _semidlt_cpu_worker(rank, world, port, n_views, case, outdir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from calibration_amd import capi
    from tests import helpers

    hm = helpers.load_hostmath()
    d, _kgt, _agt = helpers.semidlt_scene(n_views, noise=0.2, nr=2, seed=5)
    calls = []

    def allreduce(buf, count, _user):
        t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(int(count),)))
        dist.all_reduce(t)
        calls.append(int(count))
        return 0

    lo, hi, fixed = _semidlt_case(d, case)
    o = helpers.options(epsilon=1e-12)
    st, k, p, s, dist_, ve, cov = helpers.semidlt_solve_sharded(hm.hm_semidlt_solve_sharded, d, 2, o, world, rank, capi.ALLREDUCE_FN(allreduce), lo, hi, fixed)
    assert st == 0, hm.hm_semidlt_last_error()
    np.savez(os.path.join(outdir, f"sdc{rank}.npz"), k=k, p=p, dist=dist_, ve=ve, cov=cov, cost=s.final_cost, iters=s.iterations, term=s.termination,
             sizes=np.array(calls))
    dist.destroy_process_group()


def _semidlt_case(d, case):
    if case == "bounds":
        return [0.0, 0.0, 0.0, 0.0, -1.0], [float(d["kappa0"][0]) + 3.0, 4000.0, 4000.0, 4000.0, 1.0], None
    if case == "fixed":
        return None, None, [(1, 0.01)]
    return None, None, None


@pytest.mark.parametrize("n_views,world,case", [(6, 2, "plain"), (7, 2, "bounds"), (5, 3, "fixed")])
def test_semidlt_views_sharded_over_ranks_match_one_rank(hostmath, tmp_path, n_views, world, case):
    """optimize_intrinsics_semidlt (intrinsicssemidlt.cpp:155-191) with the views sharded: per evaluation the ranks exchange the 14
    pass-1 sums behind the eliminated distortion coefficients and gather the per-view table as a sum of zero-padded tables; the
    O(#views) arrow / Woodbury step then runs identically everywhere.  Same termination and iteration count as one rank, results
    equal to rounding, identical across ranks; exchange sizes are the protocol's."""
    import torch.multiprocessing as mp

    from tests import helpers

    mp.spawn(_semidlt_cpu_worker, args=(world, _free_port(), n_views, case, str(tmp_path)), nprocs=world, join=True)
    d, _kgt, _agt = helpers.semidlt_scene(n_views, noise=0.2, nr=2, seed=5)
    lo, hi, fixed = _semidlt_case(d, case)
    o = helpers.options(epsilon=1e-12)
    st, k, p, s, dist_, ve, cov = helpers.semidlt_solve(hostmath.hm_semidlt_solve, d, 2, o, lo, hi, fixed)
    assert st == 0
    res = [np.load(os.path.join(tmp_path, f"sdc{r}.npz")) for r in range(world)]
    for r in res:
        assert int(r["term"]) == s.termination and int(r["iters"]) == s.iterations
        assert set(r["sizes"].tolist()) <= {n_views, 14, n_views * (78 + 22 * 4)} and len(r["sizes"]) >= 3
        assert np.abs(r["k"] - k).max() <= 1e-9 * np.abs(k).max() and np.abs(r["p"] - p).max() <= 1e-10
        assert np.abs(r["dist"] - dist_).max() <= 1e-10 and np.abs(r["ve"] - ve).max() <= 1e-10
        assert abs(float(r["cost"]) - s.final_cost) <= 1e-10 * max(1.0, s.final_cost)
        assert np.abs(r["cov"] - cov).max() <= 1e-7 * np.abs(cov).max()
    for r in res[1:]:
        assert np.array_equal(res[0]["k"], r["k"]) and np.array_equal(res[0]["p"], r["p"]) and np.array_equal(res[0]["cov"], r["cov"])


# ---- GPU tier: the same 2-rank protocol with the REAL engine (HIP kernels per rank, both ranks on GPU 0), gloo transport --------
def _gpu_worker(rank, world, port, kind, okw, outdir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from calibration_amd import optim
    from tests import synth
    from tests import helpers

    flat = synth.shard_views(_scene(kind).flat, rank, world)
    calls = []

    def allreduce(arr):
        t = torch.from_numpy(arr)
        dist.all_reduce(t)
        calls.append(arr.size)

    o = helpers.options(epsilon=1e-12, **okw)
    with optim.ReprojHandle(flat, device=0) as h:
        h.set_allreduce(allreduce, world, rank)
        s = h.solve(o)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), intr=flat.intr, cam=flat.cam_pose if flat.cam_pose is not None else np.zeros(0),
             view=flat.view_pose if flat.view_pose is not None else np.zeros(0), iters=s.iterations, cost=s.final_cost,
             term=s.termination, ncalls=len(calls))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,okw,world", [("intr", {}, 2), ("ext", {}, 2), ("ext", {}, 4), ("ext2", dict(optimize_intrinsics=0), 3)])
def test_two_ranks_on_the_gpu_engine_match_one_rank(tmp_path, kind, okw, world):
    """2 or 4 processes, each with its own engine handle on GPU 0 and its shard of the views; the packed sum-all-reduces go
    through the host-callback transport over gloo.  (The RCCL transport needs one GPU per rank: exercised by the driver's
    N > 1 runs; 4 ranks + this process stay within the 6 GPU processes a box allows.)"""
    import torch.multiprocessing as mp

    from calibration_amd import optim
    from tests import helpers

    ctx = mp.start_processes(_gpu_worker, args=(world, _free_port(), kind, okw, str(tmp_path)), nprocs=world, join=False,
                             start_method="spawn")
    import time

    deadline = time.time() + 180
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for p in ctx.processes:
                p.kill()
            pytest.fail("2-rank GPU solve did not finish in 180 s")
    res = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    ref = _scene(kind).flat
    o = helpers.options(epsilon=1e-12, **okw)
    with optim.ReprojHandle(ref, device=0) as h:
        s = h.solve(o)
    for r in res:
        assert int(r["term"]) == s.termination and abs(int(r["iters"]) - s.iterations) <= 1 and int(r["ncalls"]) >= 2
        assert abs(float(r["cost"]) - s.final_cost) <= 1e-9 * max(1.0, s.final_cost)
        assert helpers.rel_diff(ref.intr, r["intr"]) <= 1e-9
        if ref.cam_pose is not None:
            assert helpers.rel_diff(ref.cam_pose, r["cam"]) <= 1e-9
    for r in res[1:]:
        assert np.array_equal(res[0]["intr"], r["intr"]) and np.array_equal(res[0]["cam"], r["cam"])
    views = np.concatenate([r["view"].reshape(-1, 7) for r in res])
    assert helpers.rel_diff(ref.view_pose.reshape(-1, 7), views) <= 1e-9


# ---- AX = XB over ranks: pairs partitioned by their first pose, 29 sums all-reduced -----------------------------
def _axxb_worker(rank, world, port, n_poses, outdir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from calibration_amd import optim
    from calibration_amd.geometry import pose_from_matrix
    from tests import helpers

    bTg, cTt, _X_gt, _X0 = helpers.handeye_scene(n_poses, seed=11, noise_rot_deg=0.05, noise_trans=0.0005)
    calls = []

    def allreduce(arr):
        t = torch.from_numpy(arr)
        dist.all_reduce(t)
        calls.append(arr.size)

    res = optim.optimize_handeye_sharded(bTg, cTt, allreduce, world, rank, init_gripper_se3_ref=None, min_angle_deg=1.0,
                                         options=optim.OptimOptions(epsilon=1e-12), device=0)
    np.savez(os.path.join(outdir, f"axxb{rank}.npz"), pose=pose_from_matrix(res.g_se3_c), cov=res.core.covariance,
             cost=res.core.final_cost, success=res.core.success, ncalls=len(calls), sizes=np.array(calls))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("n_poses,world", [(18, 2), (40, 3), (3, 4)])
def test_axxb_sharded_over_ranks_matches_one_gpu(tmp_path, n_poses, world):
    """estimate_and_optimize_handeye with the motion pairs split over 2 / 3 / 4 ranks (each its own process on GPU 0, sums over
    gloo; with 3 poses and 4 ranks two ranks hold no pair at all) against the single-GPU entry point."""
    import torch.multiprocessing as mp

    from calibration_amd import optim
    from calibration_amd.geometry import pose_from_matrix
    from tests import helpers

    ctx = mp.start_processes(_axxb_worker, args=(world, _free_port(), n_poses, str(tmp_path)), nprocs=world, join=False, start_method="spawn")
    import time

    deadline = time.time() + 180
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for p in ctx.processes:
                p.kill()
            pytest.fail("sharded AX = XB did not finish in 180 s")
    bTg, cTt, _X_gt, _X0 = helpers.handeye_scene(n_poses, seed=11, noise_rot_deg=0.05, noise_trans=0.0005)
    ref = optim.estimate_and_optimize_handeye(bTg, cTt, 1.0, optim.OptimOptions(epsilon=1e-12))
    ref_pose = pose_from_matrix(ref.g_se3_c)
    res = [np.load(os.path.join(tmp_path, f"axxb{r}.npz")) for r in range(world)]
    for r in res:
        assert bool(r["success"]) == ref.core.success and int(r["ncalls"]) >= 3 and set(r["sizes"].tolist()) == {29}
        assert np.abs(r["pose"] - ref_pose).max() <= 1e-10
        assert abs(float(r["cost"]) - ref.core.final_cost) <= 1e-10 * max(1.0, ref.core.final_cost)
        assert np.abs(r["cov"] - ref.core.covariance).max() <= 1e-8 * np.abs(ref.core.covariance).max()
    for r in res[1:]:  # every rank ran the same LM on the same sums
        assert np.array_equal(res[0]["pose"], r["pose"])


def _semidlt_shard(d, world, rank):
    """Contiguous view ranges balanced by count: (first_view, local views [n][4])."""
    nv = int(d["n_views"])
    v0, v1 = rank * nv // world, (rank + 1) * nv // world
    views = [np.c_[d["X"][a:b], d["Y"][a:b], d["u"][a:b], d["v"][a:b]] for a, b in zip(d["off"][v0:v1], d["off"][v0 + 1:v1 + 1])]
    return v0, views


def _semidlt_worker(rank, world, port, n_views, case, outdir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from calibration_amd import optim
    from calibration_amd.geometry import pose_from_matrix, pose_to_matrix
    from tests import helpers

    d, _kgt, _agt = helpers.semidlt_scene(n_views, rows=7, cols=9, noise=0.2, nr=2, seed=5)
    v0, views = _semidlt_shard(d, world, rank)
    calls = []

    def allreduce(arr):
        t = torch.from_numpy(arr)
        dist.all_reduce(t)
        calls.append(arr.size)

    opt = optim.IntrinsicsOptimOptions(core=optim.OptimOptions(epsilon=1e-12, compute_covariance=True), num_radial=2)
    kw = dict(bounds=optim.CalibrationBounds(fx_max=float(d["kappa0"][0]) + 3.0, fy_max=4000.0, cx_max=4000.0, cy_max=4000.0)) if case == "bounds" else \
        dict(fixed_distortion_indices=[1], fixed_distortion_values=[0.01]) if case == "fixed" else {}
    r = optim.optimize_intrinsics_semidlt_sharded(views, v0, n_views, d["kappa0"], [pose_to_matrix(p) for p in d["poses0"]], world, rank,
                                                  allreduce=allreduce, opts=opt, device=0, **kw)
    np.savez(os.path.join(outdir, f"sd{rank}.npz"), camera=r.camera, dist=r.distortion, poses=np.stack([pose_from_matrix(T) for T in r.c_se3_t]),
             ve=np.array(r.view_errors), cov=r.core.covariance if r.core.covariance is not None else np.zeros(1), cost=r.core.final_cost,
             success=r.core.success, iters=r.core.iterations, sizes=np.array(calls))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("n_views,world,case", [(9, 2, "plain"), (10, 3, "bounds"), (5, 4, "fixed"), (4, 5, "plain")])
def test_semidlt_sharded_over_ranks_matches_one_gpu(tmp_path, n_views, world, case):
    """optimize_intrinsics_semidlt with the VIEWS sharded over 2 / 3 / 4 / 5 ranks (each its own process on GPU 0, sums over gloo;
    with 4 views and 5 ranks one rank holds no view at all) against the single-GPU entry point: per evaluation one exchange of the
    m(m+1)/2 + m = 14 sums that determine the eliminated distortion coefficients and one of the per-view table (166 doubles per
    view); every rank then runs the same O(#views) step.  src/estimation/optim/intrinsicssemidlt.cpp:155-191."""
    import torch.multiprocessing as mp

    from calibration_amd import optim
    from calibration_amd.geometry import pose_from_matrix, pose_to_matrix
    from tests import helpers

    ctx = mp.start_processes(_semidlt_worker, args=(world, _free_port(), n_views, case, str(tmp_path)), nprocs=world, join=False, start_method="spawn")
    import time

    deadline = time.time() + 180
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for p in ctx.processes:
                p.kill()
            pytest.fail("sharded semi-DLT did not finish in 180 s")
    d, _kgt, _agt = helpers.semidlt_scene(n_views, rows=7, cols=9, noise=0.2, nr=2, seed=5)
    views = [np.c_[d["X"][a:b], d["Y"][a:b], d["u"][a:b], d["v"][a:b]] for a, b in zip(d["off"][:-1], d["off"][1:])]
    opt = optim.IntrinsicsOptimOptions(core=optim.OptimOptions(epsilon=1e-12, compute_covariance=True), num_radial=2)
    kw = dict(bounds=optim.CalibrationBounds(fx_max=float(d["kappa0"][0]) + 3.0, fy_max=4000.0, cx_max=4000.0, cy_max=4000.0)) if case == "bounds" else \
        dict(fixed_distortion_indices=[1], fixed_distortion_values=[0.01]) if case == "fixed" else {}
    ref = optim.optimize_intrinsics_semidlt(views, d["kappa0"], [pose_to_matrix(p) for p in d["poses0"]], opt, **kw)
    ref_poses = np.stack([pose_from_matrix(T) for T in ref.c_se3_t])
    res = [np.load(os.path.join(tmp_path, f"sd{r}.npz")) for r in range(world)]
    n2 = 78 + 22 * 4
    for r in res:
        assert bool(r["success"]) == ref.core.success and int(r["iters"]) == ref.core.iterations
        assert set(r["sizes"].tolist()) <= {n_views, 14, n_views * n2}  # view sizes once, then [N | A^T b] and the per-view table (or |r|^2 per view)
        assert np.abs(r["camera"] - ref.camera).max() <= 1e-9 * np.abs(ref.camera).max()
        assert np.abs(r["dist"] - ref.distortion).max() <= 1e-10
        assert np.abs(r["poses"] - ref_poses).max() <= 1e-10
        assert abs(float(r["cost"]) - ref.core.final_cost) <= 1e-10 * max(1.0, ref.core.final_cost)
        assert np.abs(r["ve"] - np.array(ref.view_errors)).max() <= 1e-10
        assert np.abs(r["cov"] - ref.core.covariance).max() <= 1e-7 * np.abs(ref.core.covariance).max()
        if case == "bounds":
            assert r["camera"][0] == float(d["kappa0"][0]) + 3.0
    for r in res[1:]:  # every rank ran the same solve on the same sums
        assert np.array_equal(res[0]["camera"], r["camera"]) and np.array_equal(res[0]["poses"], r["poses"])


# ---- CPU tier: many random problems over 2-4 IN-PROCESS ranks (threads; the all-reduce is a barrier + sum in Python) ------------------
def _in_process_ranks(world, body, timeout=60.0):
    """Run body(rank, reduce) on `world` threads; reduce(arr) sums a float64 array in place over the ranks (barrier + sum in a fixed
    order: bit-identical on every rank).  Every rank must issue the same sequence of all-reduces with the same sizes: a mismatch
    breaks the barrier (timeout) instead of hanging.  Returns the list of body results."""
    import threading

    barrier = threading.Barrier(world, timeout=timeout)
    slots, sizes = [None] * world, [[] for _ in range(world)]
    out, errs = [None] * world, []

    def run(rank):
        def reduce(arr):
            slots[rank] = arr.copy()
            sizes[rank].append(arr.size)
            barrier.wait()
            total = np.sum([slots[r] for r in range(world)], axis=0)
            barrier.wait()
            arr[...] = total

        try:
            out[rank] = body(rank, reduce)
        except Exception as ex:  # noqa: BLE001
            errs.append((rank, repr(ex)))
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout + 30)
    assert not errs, errs
    assert all(sizes[0] == sz for sz in sizes), "ranks issued different exchange sequences"
    return out


def _threaded_solve(hm, flats, okw):
    """Solve the shards `flats` concurrently with hm_reproj_solve_ex (the LM driver over the host-math backend); (summary, stats)."""
    from calibration_amd import capi
    from calibration_amd.capi import CbaSummary
    from tests import helpers

    world = len(flats)

    def body(rank, reduce):
        def allreduce(buf, count, _user):
            reduce(np.ctypeslib.as_array(buf, shape=(int(count),)))
            return 0

        cb = capi.ALLREDUCE_FN(allreduce)
        d = flats[rank].struct()
        s = CbaSummary()
        xs = (C.c_int64 * 8)()
        o = helpers.options(epsilon=1e-10, **okw)
        st = hm.hm_reproj_solve_ex(C.byref(d), C.byref(o), cb, None, world, rank, -1, C.byref(s), xs)
        assert st == 0, hm.hm_last_error()
        return s, [int(v) for v in xs]

    return _in_process_ranks(world, body)


def _random_case(case):
    """(kind, world, okw, make) of random problem `case`: make() builds the scene afresh (same seed: identical data)."""
    from tests import synth

    rng = np.random.default_rng(1000 + case)
    kind = ["intr", "ext", "bundle"][case % 3]
    world = int(rng.integers(2, 5))
    rough = bool(rng.integers(0, 2)) or case % 4 != 0  # most cases start far away
    okw = dict(huber_delta=float(rng.choice([1.0, -1.0, 0.3])))
    if kind != "intr":
        okw.update(optimize_intrinsics=int(rng.integers(0, 2)), optimize_extrinsics=int(rng.integers(0, 2)))
    if kind == "bundle":
        okw.update(optimize_target_pose=int(rng.integers(0, 2)))
    seed = int(rng.integers(1, 10_000))
    nv = int(rng.integers(2, 8))

    def make():
        if kind == "intr":
            sc = synth.scene_intrinsics(max(nv, 4), noise_px=0.3, seed=seed, spacing=0.06)
        elif kind == "ext":
            sc = synth.scene_extrinsics(nv, 2, noise_px=0.3, seed=seed, spacing=0.06)
        else:
            sc = synth.scene_bundle(nv + 4, 2, noise_px=0.3, seed=seed, spacing=0.04)
        if rough:  # far enough from the optimum for rejected steps, radius misses and (with bounds) line-search samples
            r2 = np.random.default_rng(case)
            sc.flat.intr[:, 5:10] = 0.0
            sc.flat.intr[:, 0:2] *= r2.uniform(0.5, 0.75)
            sc.flat.intr[:, 2:4] *= r2.uniform(0.85, 1.15, size=2)
            if sc.flat.view_pose is not None:
                sc.flat.view_pose[..., 6] *= r2.uniform(0.7, 1.4)
        return sc

    return kind, world, okw, make


@pytest.mark.parametrize("case", range(36))
def test_random_problems_over_in_process_ranks(hostmath, case):
    """The multi-rank protocol on random problems, including the rare paths: rough starts (line-search samples, rejected steps,
    radius misses, plain trials), more ranks than views (empty shards), loss on / off, intrinsics fixed / free.  Every rank must
    issue the same exchange sequence, end bit-identical in the replicated blocks and agree with the 1-rank solve."""
    from calibration_amd import capi
    from tests import synth
    from calibration_amd.capi import CbaSummary
    from tests import helpers

    _kind, world, okw, make = _random_case(case)
    ref = make().flat
    d = ref.struct()
    s1 = CbaSummary()
    o = helpers.options(epsilon=1e-10, **okw)
    assert hostmath.hm_reproj_solve(C.byref(d), C.byref(o), capi.ALLREDUCE_FN(), None, 1, 0, C.byref(s1)) == 0
    full = make().flat
    flats = [synth.shard_views(full, r, world) for r in range(world)]
    res = _threaded_solve(hostmath, flats, okw)
    for (s, xs), f in zip(res, flats):
        assert s.termination == s1.termination and abs(s.iterations - s1.iterations) <= 2, (s.report, s1.report)
        assert abs(s.final_cost - s1.final_cost) <= 1e-8 * max(1.0, s1.final_cost)
        assert np.array_equal(f.intr, flats[0].intr)
        if f.cam_pose is not None:
            assert np.array_equal(f.cam_pose, flats[0].cam_pose)
        assert helpers.rel_diff(ref.intr, f.intr) <= 1e-6
        calls, _n, spec, hits, misses, rejected, _ls, ls_evals = xs
        assert calls == 1 + s.iterations + misses + rejected + (s.successful_steps - hits - misses) + ls_evals or s.iterations == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(36))
def test_random_problems_over_in_process_ranks_gpu(case):
    """The same random problems on the REAL engine: one handle per rank on GPU 0, each driven from its own thread, the packed
    all-reduce through the host-callback transport.  Same exchange sequence on every rank, replicated blocks bit-identical,
    1-rank GPU solve reproduced."""
    from calibration_amd import optim
    from tests import synth
    from tests import helpers

    _kind, world, okw, make = _random_case(case)
    ref = make().flat
    o = helpers.options(epsilon=1e-10, **okw)
    with optim.ReprojHandle(ref, device=0) as h:
        s1 = h.solve(o)
    full = make().flat
    flats = [synth.shard_views(full, r, world) for r in range(world)]

    def body(rank, reduce):
        with optim.ReprojHandle(flats[rank], device=0) as h:
            h.set_allreduce(reduce, world, rank)
            s = h.solve(helpers.options(epsilon=1e-10, **okw))
            return s, h.solve_stats()

    res = _in_process_ranks(world, body, timeout=120.0)
    for (s, xs), f in zip(res, flats):
        assert s.termination == s1.termination and abs(s.iterations - s1.iterations) <= 2, (s.report, s1.report)
        assert abs(s.final_cost - s1.final_cost) <= 1e-8 * max(1.0, s1.final_cost)
        assert np.array_equal(f.intr, flats[0].intr)
        if f.cam_pose is not None:
            assert np.array_equal(f.cam_pose, flats[0].cam_pose)
        assert helpers.rel_diff(ref.intr, f.intr) <= 1e-6
        assert xs["allreduce_calls"] == 1 + s.iterations + xs["speculation_misses"] + xs["rejected_steps"] + xs[
            "line_search_evaluations"] + (s.successful_steps - xs["speculation_hits"] - xs["speculation_misses"]) or s.iterations == 0
