"""Synthetic calibration scenes (numpy) for bench.py and the tests.

Recipes follow the reference's test generators (tests/unit/utils.h:183-251: cumulative random robot
motions 5-25 deg / +-0.10 m, centred RxC grid, pinhole + Brown-Conrady rendering, cull Pc.z <= 1e-6)
and the configurations of BASELINE.md §3 / SURVEY.md §8(d).  The forward model here is vectorised
numpy and dtype-generic (float64 or complex128, so fixtures can be differentiated by the
complex-step method independently of both the oracle's dual numbers and the HIP kernels' analytic
Jacobians).  It is data generation only: nothing on the engine's compute path calls it.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from calibration_amd import capi
from calibration_amd.geometry import inv, make_pose, pose_from_matrix, quat_to_rotmat
from calibration_amd.optim import FlatProblem

GT_K = np.array([1000.0, 1005.0, 640.0, 360.0, 0.0])  # intrinsics_optimize_test.cpp:11-17
GT_DIST = np.array([-0.12, 0.02, 0.0005, -0.0007, 0.001])  # bundle_test.cpp:165 [k1,k2,k3,p1,p2]
GT_TAU = np.array([0.02, -0.015])  # scheimpflug_bundle_test.cpp:16-17


# ---- forward model (A.1 of SURVEY.md), dtype-generic ----------------------------------------------
def brown_conrady(x, y, k):
    """distortion.h:91-116 with coeffs [k1,k2,k3,p1,p2]."""
    r2 = x * x + y * y
    radial = 1 + k[0] * r2 + k[1] * r2 * r2 + k[2] * r2 * r2 * r2
    xd = x * radial + 2 * k[3] * x * y + k[4] * (r2 + 2 * x * x)
    yd = y * radial + k[3] * (r2 + 2 * y * y) + 2 * k[4] * x * y
    return xd, yd


def project(intr, P):
    """P: (..., 3) camera-frame points -> (..., 2) pixels; 10 params pinhole-BC, 12 Scheimpflug."""
    intr = np.asarray(intr)
    fx, fy, cx, cy, skew = intr[0], intr[1], intr[2], intr[3], intr[4]
    k = intr[5:10]
    Px, Py, Pz = P[..., 0], P[..., 1], P[..., 2]
    if intr.shape[0] == 10:  # pinhole.h:102-107
        x, y = Px / Pz, Py / Pz
        xd, yd = brown_conrady(x, y, k)
        return np.stack([fx * xd + skew * yd + cx, fy * yd + cy], axis=-1)
    tx, ty = intr[10], intr[11]  # scheimpflug.h:139-181
    ctx, stx, cty, sty = np.cos(tx), np.sin(tx), np.cos(ty), np.sin(ty)
    a = np.array([cty, 0 * cty, -sty])
    b = np.array([stx * sty, ctx, stx * cty])
    n = np.array([ctx * sty, -stx, ctx * cty])
    sden = n[0] * Px + n[1] * Py + n[2] * Pz
    mx = (a[0] * Px + a[1] * Py + a[2] * Pz) / sden
    my = (b[0] * Px + b[1] * Py + b[2] * Pz) / sden
    mx0, my0 = a[2] / n[2], b[2] / n[2]
    xd, yd = brown_conrady(mx - mx0, my - my0, k)
    return np.stack([fx * xd + skew * yd + cx + (fx * mx0 + skew * my0), fy * yd + cy + fy * my0], axis=-1)


def transform_points(T, XY):
    """(R,t) applied to planar points (X,Y,0)."""
    R, t = T[:3, :3], T[:3, 3]
    return XY[:, 0:1] * R[:, 0][None, :] + XY[:, 1:2] * R[:, 1][None, :] + t[None, :]


def make_target_grid(rows: int, cols: int, spacing: float) -> np.ndarray:
    """tests/unit/utils.h:223-231 (row-major over r then c)."""
    x0 = -0.5 * (cols - 1) * spacing
    y0 = -0.5 * (rows - 1) * spacing
    c, r = np.meshgrid(np.arange(cols), np.arange(rows))
    return np.stack([x0 + c.reshape(-1) * spacing, y0 + r.reshape(-1) * spacing], axis=1)


def rand_unit_axis(rng) -> np.ndarray:
    z = rng.uniform(-1.0, 1.0)
    t = rng.uniform(0.0, 2.0 * np.pi)
    r = np.sqrt(1.0 - z * z)
    return np.array([r * np.cos(t), r * np.sin(t), z])


def make_sequence(n_frames: int, rng) -> List[np.ndarray]:
    """b_T_g sequence of SimulatedHandEye::make_sequence (tests/unit/utils.h:203-221)."""
    T = np.eye(4)
    out = []
    for k in range(n_frames):
        out.append(T.copy())
        if k + 1 < n_frames:
            ang = np.deg2rad(rng.uniform(5.0, 25.0))
            ax = rand_unit_axis(rng)
            dt = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1)])
            T = T @ make_pose(dt, ax, ang)
    return out


def random_view_poses(n_views: int, rng, dist=2.0, max_tilt_deg=25.0, jitter=0.10, depth_spread=0.0) -> List[np.ndarray]:
    """Independent c_T_t poses: target ~`dist` m in front, tilted <= max_tilt (bench-scale scenes,
    where a cumulative random walk of 1000+ motions would leave the field of view); depth_spread > 0
    scales the distance of every view by U(1 - spread, 1 + spread)."""
    out = []
    for _ in range(n_views):
        ang = np.deg2rad(rng.uniform(5.0, max_tilt_deg))
        ax = rand_unit_axis(rng)
        tx, ty = rng.uniform(-jitter, jitter), rng.uniform(-jitter, jitter)  # (draw order x, y, z: the scenes of round 1 stay as they were)
        z = dist + rng.uniform(-jitter, jitter) if depth_spread <= 0.0 else dist * rng.uniform(1.0 - depth_spread, 1.0 + depth_spread)
        t = np.array([tx, ty, z])
        out.append(make_pose(t, ax, ang))
    return out


def render_view(intr, c_T_t, grid, noise_px=0.0, rng=None, cull=True) -> np.ndarray:
    """render_pixels (tests/unit/utils.h:233-250, culls Pc.z <= 1e-6) for one view -> PlanarView (N,4);
    cull=False is make_bundle_observations (utils.h:137-160), which projects every point."""
    Pc = transform_points(c_T_t, grid)
    keep = Pc[:, 2] > 1e-6 if cull else np.ones(len(Pc), dtype=bool)
    uv = project(intr, Pc[keep])
    if noise_px > 0 and rng is not None:
        uv = uv + rng.normal(0.0, noise_px, size=uv.shape)
    return np.concatenate([grid[keep], uv], axis=1)


def perturb_pose(T, rng, rot_deg=2.0, trans=0.01) -> np.ndarray:
    d = make_pose(rng.uniform(-trans, trans, size=3), rand_unit_axis(rng), np.deg2rad(rot_deg))
    return d @ T


def camera_gt(model: int, distortion=True) -> np.ndarray:
    base = np.concatenate([GT_K, GT_DIST if distortion else np.zeros(5)])
    return np.concatenate([base, GT_TAU]) if model == capi.CAMERA_SCHEIMPFLUG else base


def camera_init(cam_gt: np.ndarray) -> np.ndarray:
    """intrinsics_optimize_test.cpp:33-38: fx*0.97, fy*1.03, cx+5, cy-4, distortion 0 (tau 0)."""
    c = cam_gt.copy()
    c[0] *= 0.97
    c[1] *= 1.03
    c[2] += 5.0
    c[3] -= 4.0
    c[5:] = 0.0
    return c


@dataclass
class Scene:
    flat: FlatProblem
    gt_intr: np.ndarray
    gt_cam_pose: Optional[np.ndarray] = None
    gt_view_pose: Optional[np.ndarray] = None
    gt_target_pose: Optional[np.ndarray] = None
    meta: dict = field(default_factory=dict)


def scene_intrinsics(n_views=20, rows=8, cols=11, spacing=0.02, model=capi.CAMERA_PINHOLE_BC, seed=7, noise_px=0.0,
                     distortion=True, init="perturbed", first_view_global=0, tau=None, max_tilt_deg=25.0, jitter=0.10,
                     depth_spread=0.0) -> Scene:
    """C1 / C2 / C5-shaped problem: one camera, n_views views of a rows x cols grid.  tau / max_tilt_deg / jitter /
    depth_spread shape the conditioning (scene_intrinsics_wide)."""
    rng = np.random.default_rng(seed)
    cam = camera_gt(model, distortion)
    if tau is not None and model == capi.CAMERA_SCHEIMPFLUG:
        cam[10:12] = tau
    grid = make_target_grid(rows, cols, spacing)
    poses = random_view_poses(n_views, rng, max_tilt_deg=max_tilt_deg, jitter=jitter, depth_spread=depth_spread)
    views = [render_view(cam, T, grid, noise_px, rng) for T in poses]
    if init == "gt":
        cam0, poses0 = cam.copy(), poses
    else:
        cam0 = camera_init(cam)
        poses0 = [perturb_pose(T, rng) for T in poses]
    nb = len(views)
    flat = FlatProblem(capi.CHAIN_INTRINSIC, model, views, np.zeros(nb, np.int32), np.arange(nb, dtype=np.int32),
                       cam0.reshape(1, -1), None, np.stack([pose_from_matrix(T) for T in poses0]), None,
                       first_view_global=first_view_global)
    return Scene(flat, cam.reshape(1, -1), gt_view_pose=np.stack([pose_from_matrix(T) for T in poses]),
                 meta=dict(kind="intrinsics", n_views=n_views, rows=rows, cols=cols, seed=seed, noise_px=noise_px))


def scene_intrinsics_wide(n_views=20, model=capi.CAMERA_SCHEIMPFLUG, seed=3, noise_px=0.2, **kw) -> Scene:
    """A WELL-CONDITIONED intrinsics scene: a 1.3 m x 0.9 m board, tilts up to 45 degrees, distances spread over 0.6 .. 1.4 of the
    nominal 2 m, views shifted by up to 0.3 m, sensor tilt (0.2, -0.15) rad.  The default scenes follow the reference's test
    geometry (a small board, mild tilts, one distance), which leaves the Scheimpflug tilt / principal point / focal length valley
    nearly flat (condition number 1e8 .. 1e9 of the Jacobi-scaled Hessian); here the data determine every parameter and two
    correct solvers agree to rounding."""
    args = dict(rows=10, cols=14, spacing=0.1, tau=(0.2, -0.15), max_tilt_deg=45.0, jitter=0.3, depth_spread=0.4)
    args.update(kw)
    return scene_intrinsics(n_views, model=model, seed=seed, noise_px=noise_px, **args)


def ring_cameras(n_cams: int, baseline=0.25, focus=2.0) -> List[np.ndarray]:
    """c_T_r for a rig: camera 0 = identity (reference), the others offset along x by multiples of
    `baseline` on alternating sides and toed in towards the point `focus` m ahead of the reference."""
    out = [np.eye(4)]
    for c in range(1, n_cams):
        dx = baseline * ((c + 1) // 2) * (1.0 if c % 2 else -1.0)
        theta = -np.arctan2(dx, focus)
        r_T_c = make_pose(np.array([dx, 0.01 * c, 0.0]), np.array([0.0, np.sign(theta), 0.0]), abs(theta))
        out.append(inv(r_T_c))
    return out


def scene_extrinsics(n_views=8, n_cams=2, rows=8, cols=11, spacing=0.02, model=capi.CAMERA_PINHOLE_BC, seed=137,
                     noise_px=0.0, distortion=True, init="perturbed", first_view_global=0) -> Scene:
    """C3-shaped problem: every view seen by every camera."""
    rng = np.random.default_rng(seed)
    cams = []
    for _ in range(n_cams):
        cam = camera_gt(model, distortion)
        cam[0:2] *= 1 + 0.01 * rng.uniform(-1, 1, 2)
        cams.append(cam)
    grid = make_target_grid(rows, cols, spacing)
    c_T_r = ring_cameras(n_cams)
    r_T_t = random_view_poses(n_views, rng, max_tilt_deg=20.0)
    blocks, bcam, bview = [], [], []
    for v in range(n_views):
        for c in range(n_cams):
            blocks.append(render_view(cams[c], c_T_r[c] @ r_T_t[v], grid, noise_px, rng))
            bcam.append(c)
            bview.append(v)
    if init == "gt":
        cams0, cr0, rt0 = [c.copy() for c in cams], c_T_r, r_T_t
    else:
        cams0 = [camera_init(c) for c in cams]
        cr0 = [c_T_r[0]] + [perturb_pose(T, rng, 1.0, 0.01) for T in c_T_r[1:]]
        rt0 = [r_T_t[0]] + [perturb_pose(T, rng, 1.0, 0.01) for T in r_T_t[1:]]
    flat = FlatProblem(capi.CHAIN_EXTRINSIC, model, blocks, bcam, bview, np.stack(cams0),
                       np.stack([pose_from_matrix(T) for T in cr0]), np.stack([pose_from_matrix(T) for T in rt0]), None,
                       first_view_global=first_view_global)
    return Scene(flat, np.stack(cams), gt_cam_pose=np.stack([pose_from_matrix(T) for T in c_T_r]),
                 gt_view_pose=np.stack([pose_from_matrix(T) for T in r_T_t]),
                 meta=dict(kind="extrinsics", n_views=n_views, n_cams=n_cams, seed=seed))


def scene_extrinsics_shard(n_views_total, v0, v1, n_cams=8, rows=50, cols=100, spacing=0.008, model=capi.CAMERA_PINHOLE_BC, seed=137,
                           noise_px=0.2) -> Scene:
    """The views [v0, v1) of a C3-shaped problem of n_views_total views, generated WITHOUT the other views (strong-scaling
    bench: every rank builds only its shard, 5 GB for the whole of BASELINE configs[2]).  Everything shared (cameras, rig,
    their initial values) depends on `seed` only; view v draws its pose, noise and initial perturbation from the stream
    (seed, v), so the union over any partition is the same problem.  Global view 0 keeps its ground-truth pose (gauge:
    extrinsics.cpp:123-126)."""
    rng = np.random.default_rng(seed)
    cams = []
    for _ in range(n_cams):
        cam = camera_gt(model, True)
        cam[0:2] *= 1 + 0.01 * rng.uniform(-1, 1, 2)
        cams.append(cam)
    grid = make_target_grid(rows, cols, spacing)
    c_T_r = ring_cameras(n_cams)
    cams0 = [camera_init(c) for c in cams]
    cr0 = [c_T_r[0]] + [perturb_pose(T, rng, 1.0, 0.01) for T in c_T_r[1:]]
    blocks, bcam, bview, r_T_t, rt0 = [], [], [], [], []
    for v in range(v0, v1):
        vr = np.random.default_rng([seed, v])
        T = random_view_poses(1, vr, max_tilt_deg=20.0)[0]
        r_T_t.append(T)
        for c in range(n_cams):
            blocks.append(render_view(cams[c], c_T_r[c] @ T, grid, noise_px, vr))
            bcam.append(c)
            bview.append(v - v0)
        rt0.append(T if v == 0 else perturb_pose(T, vr, 1.0, 0.01))
    flat = FlatProblem(capi.CHAIN_EXTRINSIC, model, blocks, bcam, bview, np.stack(cams0),
                       np.stack([pose_from_matrix(T) for T in cr0]), np.stack([pose_from_matrix(T) for T in rt0]), None,
                       first_view_global=v0)
    return Scene(flat, np.stack(cams), gt_cam_pose=np.stack([pose_from_matrix(T) for T in c_T_r]),
                 gt_view_pose=np.stack([pose_from_matrix(T) for T in r_T_t]),
                 meta=dict(kind="extrinsics", n_views=v1 - v0, n_views_total=n_views_total, n_cams=n_cams, seed=seed))


def scene_bundle(n_poses=25, n_cams=1, rows=8, cols=11, spacing=0.02, model=capi.CAMERA_PINHOLE_BC, seed=2024,
                 noise_px=0.0, distortion=False, init="perturbed", tau=None, max_tilt_deg=25.0, jitter=0.08, depth_spread=0.0) -> Scene:
    """C4-shaped hand-eye bundle (bundle_test.cpp:9-81 recipe, n_cams cameras with small offsets).  tau / max_tilt_deg / jitter /
    depth_spread shape the conditioning as in scene_intrinsics (a WELL-CONDITIONED Scheimpflug bundle: scene_bundle_wide)."""
    rng = np.random.default_rng(seed)
    cams = [camera_gt(model, distortion) for _ in range(n_cams)]
    if tau is not None and model == capi.CAMERA_SCHEIMPFLUG:
        for cam in cams:
            cam[10:12] = tau
    g_T_c = [make_pose(np.array([0.03 + 0.05 * c, 0.01 * c, 0.12]), np.array([0.0, 1.0, 0.0]), np.deg2rad(8.0 - 3.0 * c))
             for c in range(n_cams)]
    b_T_t = make_pose(np.array([0.5, -0.1, 0.8]), np.array([1.0, 0.0, 0.0]), np.deg2rad(14.0))
    grid = make_target_grid(rows, cols, spacing)
    # robot poses: gripper looks at the target from ~1 m with random tilt
    b_T_g = []
    for _ in range(n_poses):
        c_T_t = random_view_poses(1, rng, dist=1.0, max_tilt_deg=max_tilt_deg, jitter=jitter, depth_spread=depth_spread)[0]
        b_T_g.append(b_T_t @ inv(c_T_t) @ inv(g_T_c[0]))
    blocks, bcam, btg = [], [], []
    for T in b_T_g:
        for c in range(n_cams):
            c_T_t = inv(g_T_c[c]) @ inv(T) @ b_T_t
            blocks.append(render_view(cams[c], c_T_t, grid, noise_px, rng))
            bcam.append(c)
            btg.append(np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]))
    if init == "gt":
        cams0, g0, bt0 = cams, g_T_c, b_T_t
    else:
        cams0 = [camera_init(c) for c in cams]
        g0 = [perturb_pose(T, rng, 2.0, 0.01) for T in g_T_c]
        bt0 = b_T_t
    flat = FlatProblem(capi.CHAIN_BUNDLE, model, blocks, bcam, None, np.stack(cams0),
                       np.stack([pose_from_matrix(T) for T in g0]), None, pose_from_matrix(bt0), np.stack(btg))
    return Scene(flat, np.stack(cams), gt_cam_pose=np.stack([pose_from_matrix(T) for T in g_T_c]),
                 gt_target_pose=pose_from_matrix(b_T_t), meta=dict(kind="bundle", n_poses=n_poses, n_cams=n_cams, seed=seed))


def scene_bundle_wide(n_poses=24, n_cams=2, model=capi.CAMERA_SCHEIMPFLUG, seed=3, noise_px=0.2, **kw) -> Scene:
    """A WELL-CONDITIONED hand-eye bundle for the Scheimpflug model (cf. scene_intrinsics_wide): a 0.65 m x 0.45 m board seen from
    ~1 m, tilts up to 45 degrees, distances spread over 0.6 .. 1.4 of the nominal one, sensor tilt (0.2, -0.15) rad."""
    args = dict(rows=10, cols=14, spacing=0.05, tau=(0.2, -0.15), max_tilt_deg=45.0, jitter=0.15, depth_spread=0.4, distortion=True)
    args.update(kw)
    return scene_bundle(n_poses, n_cams, model=model, seed=seed, noise_px=noise_px, **args)


def shard_views(flat: FlatProblem, rank: int, world: int) -> FlatProblem:
    """View-sharding of SURVEY.md §8(e): contiguous ranges of private views (INTRINSIC / EXTRINSIC) or of
    residual blocks (BUNDLE), balanced by observation count; shared parameters replicated."""
    nb = flat.n_blocks
    counts = np.diff(flat.blk_offset)
    if flat.chain == capi.CHAIN_BUNDLE:
        key = np.arange(nb)
        nkeys = nb
    else:
        key = flat.blk_view.astype(np.int64)
        nkeys = flat.n_views
    per_key = np.bincount(key, weights=counts, minlength=nkeys)
    cum = np.concatenate([[0], np.cumsum(per_key)])
    total = cum[-1]
    bounds = [int(np.searchsorted(cum, total * r / world, side="left")) for r in range(world + 1)]
    bounds[0], bounds[-1] = 0, nkeys
    k0, k1 = bounds[rank], bounds[rank + 1]
    sel = np.nonzero((key >= k0) & (key < k1))[0]
    views = [np.stack([flat.X[flat.blk_offset[b]:flat.blk_offset[b + 1]], flat.Y[flat.blk_offset[b]:flat.blk_offset[b + 1]],
                       flat.u[flat.blk_offset[b]:flat.blk_offset[b + 1]], flat.v[flat.blk_offset[b]:flat.blk_offset[b + 1]]], axis=1)
             for b in sel]
    if flat.chain == capi.CHAIN_BUNDLE:
        return FlatProblem(flat.chain, flat.model, views, flat.blk_cam[sel], None, flat.intr, flat.cam_pose, None,
                           flat.target_pose, flat.blk_b_T_g[sel])
    return FlatProblem(flat.chain, flat.model, views, flat.blk_cam[sel], flat.blk_view[sel] - k0, flat.intr, flat.cam_pose,
                       flat.view_pose.reshape(-1, 7)[k0:k1], None, None, first_view_global=flat.first_view_global + k0)
