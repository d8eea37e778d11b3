"""Host-side mirror of the reference's ``calib::estimation_optim`` free functions.

Same names, argument meaning, defaults and error behaviour as
``include/calib/estimation/optim/{optimize,intrinsics,extrinsics,bundle,handeye}.h`` of the
reference, expressed on numpy containers:

* ``PlanarView``         -> ``ndarray (N, 4)`` rows ``[X, Y, u, v]`` (PlanarObservation, linear/planarpose.h:22-26)
* ``MulticamPlanarView`` -> ``list[PlanarView]`` indexed by camera (linear/extrinsics.h:20)
* ``Eigen::Isometry3d``  -> ``ndarray (4, 4)``
* camera                 -> ``ndarray (10,)`` pinhole+Brown-Conrady or ``(12,)`` Scheimpflug, in
  ``CameraTraits::to_array`` order (pinhole.h:135-146, scheimpflug.h:249-260)

Everything is flattened to SoA and handed to the C ABI (``include/calibba.h``); all arithmetic on
the path runs in libcalibba's HIP kernels.  ``std::invalid_argument`` maps to ``ValueError``
(``CbaInvalidArgument``), ``std::runtime_error`` to ``CbaError``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import capi
from .capi import CbaOptions, CbaReprojProblem, CbaSummary, dptr, i32ptr, i64ptr


# ---- option / result types (optimize.h:17-40 etc.) ----------------------------------------------
@dataclass
class OptimOptions:
    optimizer: int = 0  # OptimizerType::DEFAULT
    huber_delta: float = 1.0
    epsilon: float = 1e-9
    max_iterations: int = 1000
    compute_covariance: bool = True
    verbose: bool = False


@dataclass
class OptimResult:
    success: bool = False
    covariance: Optional[np.ndarray] = None  # empty in the reference when rank deficient / disabled
    report: str = "Empty"
    final_cost: float = 0.0
    iterations: int = 0
    initial_cost: float = 0.0
    solve_seconds: float = 0.0


@dataclass
class IntrinsicsOptimOptions:  # intrinsics.h:13-20
    core: OptimOptions = field(default_factory=OptimOptions)
    num_radial: int = 2  # read by nobody in optimize_intrinsics (intrinsics.cpp:70-87), kept for parity
    optimize_skew: bool = False


@dataclass
class CalibrationBounds:  # include/calib/models/camera_matrix.h:50-72
    fx_min: float = 0.0
    fx_max: float = 2000.0
    fy_min: float = 0.0
    fy_max: float = 2000.0
    cx_min: float = 0.0
    cx_max: float = 1280.0
    cy_min: float = 0.0
    cy_max: float = 720.0
    skew_min: float = -0.01
    skew_max: float = 0.01


@dataclass
class ExtrinsicOptions:  # extrinsics.h:22-27
    core: OptimOptions = field(default_factory=OptimOptions)
    optimize_intrinsics: bool = True
    optimize_skew: bool = False
    optimize_extrinsics: bool = True


@dataclass
class BundleOptions:  # bundle.h:30-36
    core: OptimOptions = field(default_factory=OptimOptions)
    optimize_intrinsics: bool = False
    optimize_skew: bool = False
    optimize_target_pose: bool = True
    optimize_hand_eye: bool = True


@dataclass
class BundleObservation:  # bundle.h:22-26
    view: np.ndarray
    b_se3_g: np.ndarray
    camera_index: int = 0


@dataclass
class IntrinsicsOptimizationResult:  # intrinsics.h:22-28
    core: OptimResult
    camera: np.ndarray
    c_se3_t: List[np.ndarray]
    view_errors: List[float] = field(default_factory=list)  # never filled by optimize_intrinsics
    distortion: Optional[np.ndarray] = None  # optimize_intrinsics_semidlt: the fitted [k1..k_nr, p1, p2] as returned by solve_full


@dataclass
class ExtrinsicOptimizationResult:  # extrinsics.h:14-20
    core: OptimResult
    cameras: List[np.ndarray]
    c_se3_r: List[np.ndarray]
    r_se3_t: List[np.ndarray]


@dataclass
class BundleResult:  # bundle.h:39-45
    core: OptimResult
    cameras: List[np.ndarray]
    g_se3_c: List[np.ndarray]
    b_se3_t: np.ndarray


@dataclass
class PlanarPoseOptions:  # planarpose.h:12-15
    core: OptimOptions = field(default_factory=OptimOptions)
    num_radial: int = 2


@dataclass
class PlanarPoseResult:  # planarpose.h:17-22
    core: OptimResult
    pose: np.ndarray
    distortion: np.ndarray
    reprojection_error: float = 0.0


@dataclass
class OptimizeHomographyResult:  # homography.h:12-15
    core: OptimResult
    homography: np.ndarray


@dataclass
class HandeyeResult:  # handeye.h:16-19
    core: OptimResult
    g_se3_c: np.ndarray


# ---- pose helpers (observationutils.h:43-62): host logic, see geometry.py -----------------------
from .geometry import pose_from_matrix, pose_to_matrix, poses_from_matrices, poses_to_matrices  # noqa: E402


def camera_model_of(cam: np.ndarray) -> int:
    n = int(np.asarray(cam).shape[-1])
    if n == 10:
        return capi.CAMERA_PINHOLE_BC
    if n == 12:
        return capi.CAMERA_SCHEIMPFLUG
    raise ValueError(f"camera parameter vector must have 10 or 12 entries, got {n}")


def to_cba_options(core: OptimOptions, optimize_intrinsics=True, optimize_skew=False, optimize_extrinsics=True,
                   optimize_target_pose=True) -> CbaOptions:
    o = CbaOptions()
    o.optimizer = int(core.optimizer)
    o.max_iterations = int(core.max_iterations)
    o.huber_delta = float(core.huber_delta)
    o.epsilon = float(core.epsilon)
    o.compute_covariance = int(bool(core.compute_covariance))
    o.verbose = int(bool(core.verbose))
    o.optimize_intrinsics = int(bool(optimize_intrinsics))
    o.optimize_skew = int(bool(optimize_skew))
    o.optimize_extrinsics = int(bool(optimize_extrinsics))
    o.optimize_target_pose = int(bool(optimize_target_pose))
    return o


def result_core(s: CbaSummary, cov: Optional[np.ndarray]) -> OptimResult:
    return OptimResult(
        success=bool(s.success),
        covariance=cov,
        report=s.report.decode("utf-8", "replace"),
        final_cost=float(s.final_cost),
        iterations=int(s.iterations),
        initial_cost=float(s.initial_cost),
        solve_seconds=float(s.solve_seconds),
    )


# ---- flattening: AoS host containers -> SoA + CSR ------------------------
class FlatProblem:
    """Owns the numpy buffers a ``cba_reproj_problem`` points into."""

    def __init__(self, chain: int, model: int, views: Sequence[np.ndarray], blk_cam, blk_view, intr, cam_pose,
                 view_pose, target_pose, blk_b_T_g=None, first_view_global: int = 0):
        self.chain, self.model = chain, model
        counts = [int(np.asarray(v).reshape(-1, 4).shape[0]) if np.asarray(v).size else 0 for v in views]
        self.blk_offset = np.zeros(len(views) + 1, dtype=np.int64)
        np.cumsum(counts, out=self.blk_offset[1:])
        if len(views) and self.blk_offset[-1] > 0:
            allv = np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1, 4) for v in views], axis=0)
        else:
            allv = np.zeros((0, 4))
        self.X = np.ascontiguousarray(allv[:, 0])
        self.Y = np.ascontiguousarray(allv[:, 1])
        self.u = np.ascontiguousarray(allv[:, 2])
        self.v = np.ascontiguousarray(allv[:, 3])
        self.blk_cam = np.ascontiguousarray(blk_cam, dtype=np.int32)
        self.blk_view = None if blk_view is None else np.ascontiguousarray(blk_view, dtype=np.int32)
        self.blk_b_T_g = None if blk_b_T_g is None else np.ascontiguousarray(blk_b_T_g, dtype=np.float64)
        self.intr = np.ascontiguousarray(intr, dtype=np.float64).copy()
        self.cam_pose = None if cam_pose is None else np.ascontiguousarray(cam_pose, dtype=np.float64).copy()
        self.view_pose = None if view_pose is None else np.ascontiguousarray(view_pose, dtype=np.float64).copy()
        self.target_pose = None if target_pose is None else np.ascontiguousarray(target_pose, dtype=np.float64).copy()
        self.n_blocks = len(views)
        self.n_cams = int(self.intr.reshape(-1, capi_intr_size(model)).shape[0]) if self.intr.size else 0
        self.n_views = 0 if self.view_pose is None else int(self.view_pose.reshape(-1, 7).shape[0])
        self.first_view_global = int(first_view_global)

    @property
    def n_obs(self) -> int:
        return int(self.blk_offset[-1])

    def struct(self) -> CbaReprojProblem:
        d = CbaReprojProblem()
        d.chain, d.camera_model = self.chain, self.model
        d.n_blocks, d.n_cams, d.n_views = self.n_blocks, self.n_cams, self.n_views
        d.first_view_global = self.first_view_global
        d.blk_offset = i64ptr(self.blk_offset)
        d.blk_cam = i32ptr(self.blk_cam)
        d.blk_view = i32ptr(self.blk_view)
        d.blk_b_T_g = dptr(self.blk_b_T_g)
        d.X, d.Y, d.u, d.v = dptr(self.X), dptr(self.Y), dptr(self.u), dptr(self.v)
        d.intr = dptr(self.intr)
        d.cam_pose = dptr(self.cam_pose)
        d.view_pose = dptr(self.view_pose)
        d.target_pose = dptr(self.target_pose)
        return d


def capi_intr_size(model: int) -> int:
    return 12 if model == capi.CAMERA_SCHEIMPFLUG else 10


def matrix_to_rt12(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64).reshape(4, 4)
    return np.concatenate([m[:3, :3].reshape(-1), m[:3, 3]])


def flatten_intrinsics(views, init_camera, init_c_se3_t) -> FlatProblem:
    model = camera_model_of(init_camera)
    poses = np.stack([pose_from_matrix(m) for m in init_c_se3_t]) if len(init_c_se3_t) else np.zeros((0, 7))
    nb = len(views)
    return FlatProblem(capi.CHAIN_INTRINSIC, model, views, np.zeros(nb, np.int32), np.arange(nb, dtype=np.int32),
                       np.asarray(init_camera, dtype=np.float64).reshape(1, -1), None, poses, None)


def flatten_extrinsics(views, init_cameras, init_c_se3_r, init_r_se3_t) -> FlatProblem:
    model = camera_model_of(init_cameras[0])
    blocks, bcam, bview = [], [], []
    for vi, mv in enumerate(views):
        for ci in range(len(init_cameras)):
            pv = np.asarray(mv[ci]).reshape(-1, 4) if ci < len(mv) else np.zeros((0, 4))
            if pv.shape[0] == 0:  # extrinsics.cpp:94-96: empty per-camera view skipped
                continue
            blocks.append(pv)
            bcam.append(ci)
            bview.append(vi)
    cams = np.stack([pose_from_matrix(m) for m in init_c_se3_r])
    tgts = np.stack([pose_from_matrix(m) for m in init_r_se3_t]) if len(init_r_se3_t) else np.zeros((0, 7))
    return FlatProblem(capi.CHAIN_EXTRINSIC, model, blocks, bcam, bview, np.stack(init_cameras), cams, tgts, None)


def flatten_bundle(observations: Sequence[BundleObservation], cameras, init_g_se3_c, init_b_se3_t) -> FlatProblem:
    model = camera_model_of(cameras[0]) if len(cameras) else capi.CAMERA_PINHOLE_BC
    blocks = [np.asarray(o.view, dtype=np.float64).reshape(-1, 4) for o in observations]
    bcam = [int(o.camera_index) for o in observations]
    btg = np.stack([matrix_to_rt12(o.b_se3_g) for o in observations]) if len(observations) else np.zeros((0, 12))
    g = np.stack([pose_from_matrix(m) for m in init_g_se3_c]) if len(init_g_se3_c) else np.zeros((0, 7))
    intr = np.stack(cameras) if len(cameras) else np.zeros((0, 10))
    return FlatProblem(capi.CHAIN_BUNDLE, model, blocks, bcam, None, intr, g, None, pose_from_matrix(init_b_se3_t), btg)


# ---- handle wrapper ------------------------------------------------------------------------------
class ReprojHandle:
    """RAII wrapper of ``cba_reproj``: observations + parameters resident in HBM."""

    def __init__(self, flat: FlatProblem, device: int = 0, lib=None, records: Optional[Sequence[np.ndarray]] = None):
        """``records`` (optional): one C-contiguous float64 array of shape (n_b, 4) per residual block holding its
        observations as {object_x, object_y, image_u, image_v} rows — the layout of the reference's
        ``std::vector<PlanarObservation>`` — handed to ``cba_reproj_create_aos`` and read in place (the flat X, Y, u, v
        arrays are then not used)."""
        self.lib = lib or capi.load_library()
        self.flat = flat
        self._desc = flat.struct()
        h = C.c_void_p()
        if records is None:
            capi.check(self.lib, self.lib.cba_reproj_create(C.byref(self._desc), int(device), C.byref(h)))
        else:
            if len(records) != flat.n_blocks:
                raise ValueError("one record array per residual block")
            recs = [np.ascontiguousarray(r, dtype=np.float64).reshape(-1, 4) for r in records]
            for b, r in enumerate(recs):
                if r.shape[0] != int(flat.blk_offset[b + 1] - flat.blk_offset[b]):
                    raise ValueError("record count differs from blk_offset")
            ptrs = (capi.c_double_p * max(1, len(recs)))(*[capi.dptr(r) for r in recs])
            self._desc.X = self._desc.Y = self._desc.u = self._desc.v = None
            capi.check(self.lib, self.lib.cba_reproj_create_aos(C.byref(self._desc), ptrs, int(device), C.byref(h)))
        self.h = h
        self._cb = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.cba_reproj_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def n_obs(self) -> int:
        return int(self.lib.cba_reproj_num_observations(self.h))

    @property
    def local_columns(self) -> int:
        return int(self.lib.cba_local_columns(self.flat.chain, self.flat.model))

    def set_params(self, intr=None, cam_pose=None, view_pose=None, target_pose=None):
        f = self.flat
        for name, val in (("intr", intr), ("cam_pose", cam_pose), ("view_pose", view_pose), ("target_pose", target_pose)):
            if val is not None:
                getattr(f, name)[...] = np.asarray(val, dtype=np.float64).reshape(getattr(f, name).shape)
        capi.check(self.lib, self.lib.cba_reproj_set_params(self.h, dptr(f.intr), dptr(f.cam_pose), dptr(f.view_pose),
                                                            dptr(f.target_pose)))

    def get_params(self):
        f = self.flat
        capi.check(self.lib, self.lib.cba_reproj_get_params(self.h, dptr(f.intr), dptr(f.cam_pose), dptr(f.view_pose),
                                                            dptr(f.target_pose)))
        return f.intr, f.cam_pose, f.view_pose, f.target_pose

    def eval(self):
        capi.check(self.lib, self.lib.cba_reproj_eval(self.h))

    def eval_fetch(self):
        n, p = self.n_obs, self.local_columns
        r = np.zeros(2 * n)
        J = np.zeros((2 * n, p))
        capi.check(self.lib, self.lib.cba_reproj_eval_fetch(self.h, dptr(r), dptr(J)))
        return r, J

    def eval_fetch_blocks(self, b0: int, b1: int):
        """Residuals and Jacobian rows of the residual blocks [b0, b1) only (cba_reproj_eval_fetch_blocks)."""
        n, p = int(self.flat.blk_offset[b1] - self.flat.blk_offset[b0]), self.local_columns
        r = np.zeros(2 * n)
        J = np.zeros((2 * n, p))
        capi.check(self.lib, self.lib.cba_reproj_eval_fetch_blocks(self.h, int(b0), int(b1), dptr(r), dptr(J)))
        return r, J

    def set_scalar(self, scalar: int):
        """0 = fp64 per-observation arithmetic (default), 1 = fp32 (accumulators stay fp64)."""
        capi.check(self.lib, self.lib.cba_reproj_set_scalar(self.h, int(scalar)))

    def eval_fetch_f32(self):
        n, p = self.n_obs, self.local_columns
        r = np.zeros(2 * n, dtype=np.float32)
        J = np.zeros((2 * n, p), dtype=np.float32)
        capi.check(self.lib, self.lib.cba_reproj_eval_fetch_f32(self.h, r.ctypes.data_as(C.POINTER(C.c_float)),
                                                                J.ctypes.data_as(C.POINTER(C.c_float))))
        return r, J

    def eval_timed(self, warmup: int, iters: int) -> float:
        ms = C.c_double(0.0)
        capi.check(self.lib, self.lib.cba_reproj_eval_timed(self.h, int(warmup), int(iters), C.byref(ms)))
        return float(ms.value)

    def normal_eq_timed(self, warmup: int, iters: int) -> float:
        """Average milliseconds of one Mode B pass (per-block normal equations), HIP events on the engine's stream."""
        ms = C.c_double()
        capi.check(self.lib, self.lib.cba_reproj_normal_eq_timed(self.h, int(warmup), int(iters), C.byref(ms)))
        return ms.value

    def cost(self, huber_delta: float = 1.0) -> float:
        c = C.c_double(0.0)
        capi.check(self.lib, self.lib.cba_reproj_cost(self.h, float(huber_delta), C.byref(c)))
        return float(c.value)

    def block_normal_eq(self) -> np.ndarray:
        w = int(self.lib.cba_reproj_block_normal_eq_size(self.h))
        out = np.zeros((self.flat.n_blocks, w))
        capi.check(self.lib, self.lib.cba_reproj_block_normal_eq(self.h, dptr(out)))
        return out

    def set_lm_mode(self, mode: int):
        """0 = staged LM iteration (device controller), 1 = automatic (default), 2 = the resident single-launch kernel whenever it
        can take the problem, 3 = staged with the reduced solve and step decision on the host (diagnostic A/B); cba_reproj_set_lm_mode."""
        capi.check(self.lib, self.lib.cba_reproj_set_lm_mode(self.h, int(mode)))

    def solve_stats(self) -> dict:
        """What the last host-driven solve exchanged between ranks (cba_reproj_solve_stats)."""
        a = (C.c_int64 * 8)()
        capi.check(self.lib, self.lib.cba_reproj_solve_stats(self.h, a))
        return dict(zip(("allreduce_calls", "allreduce_doubles", "speculative_steps", "speculation_hits", "speculation_misses",
                         "rejected_steps", "line_searches", "line_search_evaluations"), (int(v) for v in a)))

    def solve(self, opts: CbaOptions) -> CbaSummary:
        s = CbaSummary()
        capi.check(self.lib, self.lib.cba_reproj_solve(self.h, C.byref(opts), C.byref(s)))
        self.get_params()
        return s

    def covariance(self, opts: CbaOptions) -> Optional[np.ndarray]:
        dim = int(self.lib.cba_reproj_covariance_dim(self.h))
        cov = np.zeros((dim, dim))
        st = self.lib.cba_reproj_covariance(self.h, C.byref(opts), dptr(cov))
        if st == capi.CBA_ERR_RUNTIME:  # rank deficient: the reference leaves the matrix empty
            return None
        capi.check(self.lib, st)
        return cov

    def covariance_shared(self, opts: CbaOptions) -> Optional[np.ndarray]:
        """Marginal covariance of the shared blocks only (cba_reproj_covariance_shared): O(#views) work."""
        n = int(self.lib.cba_reproj_covariance_shared_dim(self.h))
        cov = np.zeros((n, n))
        st = self.lib.cba_reproj_covariance_shared(self.h, C.byref(opts), dptr(cov))
        if st == capi.CBA_ERR_RUNTIME:
            return None
        capi.check(self.lib, st)
        return cov

    def covariance_views(self, opts: CbaOptions, view_idx) -> Optional[np.ndarray]:
        """Marginal 7 x 7 covariance [quaternion, translation] of the poses of the listed views (cba_reproj_covariance_views):
        the diagonal blocks of the reference-layout matrix without forming it.  None if rank deficient."""
        idx = np.ascontiguousarray(view_idx, dtype=np.int32)
        out = np.zeros((len(idx), 7, 7))
        st = self.lib.cba_reproj_covariance_views(self.h, C.byref(opts), len(idx), i32ptr(idx), dptr(out))
        if st == capi.CBA_ERR_RUNTIME:
            return None
        capi.check(self.lib, st)
        return out

    def set_allreduce(self, fn, n_ranks: int, rank: int):
        """fn(np.ndarray) sums the array in place across ranks (host buffers)."""

        def _cb(buf, count, _user):
            try:
                arr = np.ctypeslib.as_array(buf, shape=(int(count),))
                fn(arr)
                return 0
            except Exception:  # pragma: no cover
                return 1

        self._cb = capi.ALLREDUCE_FN(_cb)
        capi.check(self.lib, self.lib.cba_reproj_set_allreduce(self.h, self._cb, None, int(n_ranks), int(rank)))

    def init_rccl(self, unique_id: bytes, n_ranks: int, rank: int):
        buf = (C.c_uint8 * capi.RCCL_UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        capi.check(self.lib, self.lib.cba_reproj_init_rccl(self.h, buf, int(n_ranks), int(rank)))


def rccl_unique_id(lib=None) -> bytes:
    lib = lib or capi.load_library()
    buf = (C.c_uint8 * capi.RCCL_UNIQUE_ID_BYTES)()
    capi.check(lib, lib.cba_rccl_unique_id(buf))
    return bytes(buf)


def _solve_flat(flat: FlatProblem, copts: CbaOptions, device: int = 0):
    with ReprojHandle(flat, device) as h:
        s = h.solve(copts)
        cov = h.covariance(copts) if copts.compute_covariance else None
    return s, cov


# ---- the reference's free functions ---------------------------------------------------------------
def optimize_intrinsics(views, init_camera, init_c_se3_t, opts: Optional[IntrinsicsOptimOptions] = None,
                        device: int = 0) -> IntrinsicsOptimizationResult:
    """optimize_intrinsics<CameraT> (intrinsics.h:35-39, intrinsics.cpp:98-120)."""
    opts = opts or IntrinsicsOptimOptions()
    if len(views) < 4:  # intrinsics.cpp:92-96
        raise capi.CbaInvalidArgument(capi.CBA_ERR_INVALID_ARGUMENT,
                                      "Insufficient views for calibration (at least 4 required).")
    flat = flatten_intrinsics(views, init_camera, init_c_se3_t)
    s, cov = _solve_flat(flat, to_cba_options(opts.core, True, opts.optimize_skew), device)
    return IntrinsicsOptimizationResult(result_core(s, cov), flat.intr.reshape(-1).copy(),
                                        [pose_to_matrix(p) for p in flat.view_pose.reshape(-1, 7)])


def optimize_extrinsics(views, init_cameras, init_c_se3_r, init_r_se3_t, opts: Optional[ExtrinsicOptions] = None,
                        device: int = 0) -> ExtrinsicOptimizationResult:
    """optimize_extrinsics<CameraT> (extrinsics.h:29-34, extrinsics.cpp:174-196)."""
    opts = opts or ExtrinsicOptions()
    if len(init_c_se3_r) != len(init_cameras) or len(init_r_se3_t) != len(views):  # extrinsics.cpp:162-172
        raise capi.CbaInvalidArgument(capi.CBA_ERR_INVALID_ARGUMENT,
                                      "Incompatible pose vector sizes for joint optimization")
    flat = flatten_extrinsics(views, init_cameras, init_c_se3_r, init_r_se3_t)
    s, cov = _solve_flat(flat, to_cba_options(opts.core, opts.optimize_intrinsics, opts.optimize_skew,
                                              opts.optimize_extrinsics), device)
    P = capi_intr_size(flat.model)
    return ExtrinsicOptimizationResult(result_core(s, cov), [c.copy() for c in flat.intr.reshape(-1, P)],
                                       [pose_to_matrix(p) for p in flat.cam_pose.reshape(-1, 7)],
                                       [pose_to_matrix(p) for p in flat.view_pose.reshape(-1, 7)])


def optimize_bundle(observations, initial_cameras, init_g_se3_c, init_b_se3_t, opts: Optional[BundleOptions] = None,
                    device: int = 0) -> BundleResult:
    """optimize_bundle<CameraT> (bundle.h:58-63, bundle.cpp:147-170)."""
    opts = opts or BundleOptions()
    if len(initial_cameras) == 0:  # bundle.cpp:139-141
        raise capi.CbaInvalidArgument(capi.CBA_ERR_INVALID_ARGUMENT, "No camera intrinsics provided")
    if len(observations) == 0:  # bundle.cpp:142-144
        raise capi.CbaInvalidArgument(capi.CBA_ERR_INVALID_ARGUMENT, "No observations provided")
    flat = flatten_bundle(observations, initial_cameras, init_g_se3_c, init_b_se3_t)
    s, cov = _solve_flat(flat, to_cba_options(opts.core, opts.optimize_intrinsics, opts.optimize_skew,
                                              opts.optimize_hand_eye, opts.optimize_target_pose), device)
    P = capi_intr_size(flat.model)
    return BundleResult(result_core(s, cov), [c.copy() for c in flat.intr.reshape(-1, P)],
                        [pose_to_matrix(p) for p in flat.cam_pose.reshape(-1, 7)], pose_to_matrix(flat.target_pose))


def optimize_handeye(base_se3_gripper, camera_se3_target, init_gripper_se3_ref, options: Optional[OptimOptions] = None
                     ) -> HandeyeResult:
    """optimize_handeye (handeye.h:40-43, handeye.cpp:60-78)."""
    options = options or OptimOptions()
    lib = capi.load_library()
    n = len(base_se3_gripper)
    bg = np.stack([pose_from_matrix(m) for m in base_se3_gripper]) if n else np.zeros((0, 7))
    ct = np.stack([pose_from_matrix(m) for m in camera_se3_target]) if len(camera_se3_target) else np.zeros((0, 7))
    if len(camera_se3_target) != n:
        raise capi.CbaError(capi.CBA_ERR_RUNTIME, "Inconsistent hand-eye input sizes")
    x = pose_from_matrix(init_gripper_se3_ref)
    copts = to_cba_options(options)
    s = CbaSummary()
    cov = np.zeros((7, 7))
    capi.check(lib, lib.cba_optimize_handeye(n, dptr(bg), dptr(ct), dptr(x), C.byref(copts), C.byref(s),
                                             dptr(cov) if options.compute_covariance else dptr(None)))
    return HandeyeResult(result_core(s, cov if options.compute_covariance else None), pose_to_matrix(x))


def optimize_handeye_sharded(base_se3_gripper, camera_se3_target, allreduce, n_ranks: int, rank: int, init_gripper_se3_ref=None,
                             min_angle_deg: float = 1.0, options: Optional[OptimOptions] = None, device: int = 0) -> HandeyeResult:
    """optimize_handeye / estimate_and_optimize_handeye on several GPUs (cba_estimate_and_optimize_handeye_sharded): every rank
    passes all poses and evaluates its share of the motion pairs; ``allreduce(np.ndarray)`` sums the 29 accumulated values in
    place across ranks.  ``init_gripper_se3_ref=None`` starts from the all-pairs Tsai-Lenz seed."""
    options = options or OptimOptions()
    lib = capi.load_library()
    n = len(base_se3_gripper)
    if len(camera_se3_target) != n:
        raise capi.CbaError(capi.CBA_ERR_RUNTIME, "Inconsistent hand-eye input sizes")
    bg = np.stack([pose_from_matrix(m) for m in base_se3_gripper]) if n else np.zeros((0, 7))
    ct = np.stack([pose_from_matrix(m) for m in camera_se3_target]) if n else np.zeros((0, 7))
    x = np.zeros(7) if init_gripper_se3_ref is None else pose_from_matrix(init_gripper_se3_ref)

    def _cb(buf, count, _user):
        try:
            allreduce(np.ctypeslib.as_array(buf, shape=(int(count),)))
            return 0
        except Exception:  # pragma: no cover
            return 1

    cb = capi.ALLREDUCE_FN(_cb)
    copts = to_cba_options(options)
    s = CbaSummary()
    cov = np.zeros((7, 7))
    capi.check(lib, lib.cba_estimate_and_optimize_handeye_sharded(
        n, dptr(bg), dptr(ct), float(min_angle_deg), 1 if init_gripper_se3_ref is None else 0, dptr(x), C.byref(copts), C.byref(s),
        dptr(cov) if options.compute_covariance else dptr(None), cb, None, int(n_ranks), int(rank), int(device)))
    return HandeyeResult(result_core(s, cov if options.compute_covariance else None), pose_to_matrix(x))


def optimize_handeye_rccl(base_se3_gripper, camera_se3_target, rccl_id: bytes, n_ranks: int, rank: int, init_gripper_se3_ref=None,
                          min_angle_deg: float = 1.0, options: Optional[OptimOptions] = None, device: int = 0) -> HandeyeResult:
    """The sharded AX = XB solve with RCCL over xGMI as the transport (cba_estimate_and_optimize_handeye_rccl): ``rccl_id`` is the
    128-byte id of rccl_unique_id(), created by one rank and distributed by the caller; every rank calls this (a collective)."""
    options = options or OptimOptions()
    lib = capi.load_library()
    n = len(base_se3_gripper)
    if len(camera_se3_target) != n:
        raise capi.CbaError(capi.CBA_ERR_RUNTIME, "Inconsistent hand-eye input sizes")
    bg = np.stack([pose_from_matrix(m) for m in base_se3_gripper]) if n else np.zeros((0, 7))
    ct = np.stack([pose_from_matrix(m) for m in camera_se3_target]) if n else np.zeros((0, 7))
    x = np.zeros(7) if init_gripper_se3_ref is None else pose_from_matrix(init_gripper_se3_ref)
    copts = to_cba_options(options)
    s = CbaSummary()
    cov = np.zeros((7, 7))
    idbuf = (C.c_uint8 * capi.RCCL_UNIQUE_ID_BYTES).from_buffer_copy(bytes(rccl_id))
    capi.check(lib, lib.cba_estimate_and_optimize_handeye_rccl(
        n, dptr(bg), dptr(ct), float(min_angle_deg), 1 if init_gripper_se3_ref is None else 0, dptr(x), C.byref(copts), C.byref(s),
        dptr(cov) if options.compute_covariance else dptr(None), idbuf, int(n_ranks), int(rank), int(device)))
    return HandeyeResult(result_core(s, cov if options.compute_covariance else None), pose_to_matrix(x))


def estimate_handeye_dlt(base_se3_gripper, camera_se3_target, min_angle_deg: float = 1.0) -> np.ndarray:
    """estimate_handeye_dlt (linear/handeye.h, handeyedlt.cpp:126-137): all-pairs Tsai-Lenz seed, pairs enumerated on the GPU."""
    lib = capi.load_library()
    n = len(base_se3_gripper)
    if len(camera_se3_target) != n:
        raise capi.CbaError(capi.CBA_ERR_RUNTIME, "Inconsistent hand-eye input sizes")
    bg = np.stack([pose_from_matrix(m) for m in base_se3_gripper]) if n else np.zeros((0, 7))
    ct = np.stack([pose_from_matrix(m) for m in camera_se3_target]) if n else np.zeros((0, 7))
    x = np.zeros(7)
    capi.check(lib, lib.cba_estimate_handeye_dlt(n, dptr(bg), dptr(ct), float(min_angle_deg), dptr(x)))
    return pose_to_matrix(x)


def estimate_and_optimize_handeye(base_se3_gripper, camera_se3_target, min_angle_deg: float = 1.0,
                                  options: Optional[OptimOptions] = None) -> HandeyeResult:
    """estimate_and_optimize_handeye (optim/handeye.h:64-67, handeye.cpp:80-87)."""
    init = estimate_handeye_dlt(base_se3_gripper, camera_se3_target, min_angle_deg)
    return optimize_handeye(base_se3_gripper, camera_se3_target, init, options)


def optimize_planar_pose_batch(views, intrinsics, init_poses, opts: Optional[PlanarPoseOptions] = None) -> List[PlanarPoseResult]:
    """Batched optimize_planar_pose: every view is an independent 6-parameter variable-projection solve, all run
    in ONE kernel launch (one GPU thread per view).  intrinsics = [fx, fy, cx, cy, skew]."""
    opts = opts or PlanarPoseOptions()
    lib = capi.load_library()
    nv = len(views)
    vs = [np.asarray(v, dtype=np.float64).reshape(-1, 4) for v in views]
    off = np.zeros(nv + 1, dtype=np.int64)
    np.cumsum([v.shape[0] for v in vs], out=off[1:])
    allv = np.concatenate(vs, axis=0) if nv else np.zeros((0, 4))
    X, Y, u, v = (np.ascontiguousarray(allv[:, k]) for k in range(4))
    K = np.ascontiguousarray(np.asarray(intrinsics, dtype=np.float64).reshape(5))
    poses = np.ascontiguousarray(poses_from_matrices(np.asarray(init_poses, dtype=np.float64))) if nv else np.zeros((0, 7))
    m = int(opts.num_radial) + 2
    summ = (CbaSummary * max(nv, 1))()
    dist = np.zeros((max(nv, 1), m))
    rms = np.zeros(max(nv, 1))
    cov = np.zeros((max(nv, 1), 36))
    copts = to_cba_options(opts.core)
    capi.check(lib, lib.cba_optimize_planar_pose_batch(nv, i64ptr(off), dptr(X), dptr(Y), dptr(u), dptr(v), dptr(K), int(opts.num_radial),
                                                       dptr(poses), C.byref(copts), summ, dptr(dist), dptr(rms),
                                                       dptr(cov) if opts.core.compute_covariance else dptr(None)))
    out = []
    mats = poses_to_matrices(poses) if nv else np.zeros((0, 4, 4))
    has_cov = np.any(cov != 0.0, axis=1) if opts.core.compute_covariance else np.zeros(max(nv, 1), dtype=bool)
    for i in range(nv):
        out.append(PlanarPoseResult(result_core(summ[i], cov[i].reshape(6, 6).copy() if has_cov[i] else None), mats[i], dist[i].copy(), float(rms[i])))
    return out


def optimize_planar_pose(view, intrinsics, init_pose, opts: Optional[PlanarPoseOptions] = None) -> PlanarPoseResult:
    """optimize_planar_pose (planarpose.h:24-26, planarpose.cpp:84-127)."""
    return optimize_planar_pose_batch([view], intrinsics, [init_pose], opts)[0]


def optimize_homography_batch(views, init_hs, options: Optional[OptimOptions] = None) -> List[OptimizeHomographyResult]:
    """Batched optimize_homography: every view is an independent 8-parameter refinement (one residual block and one
    Huber loss per correspondence), all views in ONE kernel launch (one wavefront per view)."""
    options = options or OptimOptions()
    lib = capi.load_library()
    nv = len(views)
    vs = [np.asarray(v, dtype=np.float64).reshape(-1, 4) for v in views]
    off = np.zeros(nv + 1, dtype=np.int64)
    np.cumsum([v.shape[0] for v in vs], out=off[1:])
    allv = np.concatenate(vs, axis=0) if nv else np.zeros((0, 4))
    X, Y, u, v = (np.ascontiguousarray(allv[:, k]) for k in range(4))
    H = np.ascontiguousarray(np.stack([np.asarray(h, dtype=np.float64).reshape(9) for h in init_hs])) if nv else np.zeros((0, 9))
    summ = (CbaSummary * max(nv, 1))()
    cov = np.zeros((max(nv, 1), 64))
    copts = to_cba_options(options)
    capi.check(lib, lib.cba_optimize_homography_batch(nv, i64ptr(off), dptr(X), dptr(Y), dptr(u), dptr(v), dptr(H), C.byref(copts), summ,
                                                      dptr(cov) if options.compute_covariance else dptr(None)))
    out = []
    for i in range(nv):
        c = cov[i].reshape(8, 8).copy() if options.compute_covariance and np.any(cov[i]) else None
        out.append(OptimizeHomographyResult(result_core(summ[i], c), H[i].reshape(3, 3).copy()))
    return out


def optimize_homography(data, init_h, options: Optional[OptimOptions] = None) -> OptimizeHomographyResult:
    """optimize_homography (homography.h:17-18, homography.cpp:144-175); data rows = [x, y, u, v]."""
    return optimize_homography_batch([data], [init_h], options)[0]


def optimize_intrinsics_semidlt(views, initial_guess, init_c_se3_t=None, opts: Optional[IntrinsicsOptimOptions] = None,
                                bounds: Optional[CalibrationBounds] = None, fixed_distortion_indices=(), fixed_distortion_values=()
                                ) -> IntrinsicsOptimizationResult:
    """optimize_intrinsics_semidlt (intrinsics.h:30-33, intrinsicssemidlt.cpp:155-191).  initial_guess = [fx, fy, cx, cy, skew].
    init_c_se3_t: the per-view seeds the reference computes inside the call with calib::estimate_planar_pose (:37-40);
    None = compute them the same way with the batched GPU seed (cba_estimate_planar_pose_batch)."""
    opts = opts or IntrinsicsOptimOptions()
    lib = capi.load_library()
    nv = len(views)
    vs = [np.asarray(v, dtype=np.float64).reshape(-1, 4) for v in views]
    off = np.zeros(nv + 1, dtype=np.int64)
    np.cumsum([v.shape[0] for v in vs], out=off[1:])
    allv = np.concatenate(vs, axis=0) if nv else np.zeros((0, 4))
    X, Y, u, v = (np.ascontiguousarray(allv[:, k]) for k in range(4))
    K = np.ascontiguousarray(np.asarray(initial_guess, dtype=np.float64).reshape(5)).copy()
    poses = np.ascontiguousarray(np.stack([pose_from_matrix(T) for T in init_c_se3_t])) if nv and init_c_se3_t is not None else np.zeros((max(nv, 1), 7))
    nr = int(getattr(opts, "num_radial", 2))
    if init_c_se3_t is None and nv >= 4:  # IntrinsicBlocks::create (:37-40): the batched GPU seed
        poses = np.ascontiguousarray(np.stack([pose_from_matrix(T) for T in estimate_planar_pose_batch(views, K)]))
    copts = to_cba_options(opts.core, optimize_skew=opts.optimize_skew)
    s = CbaSummary()
    dist, ve = np.zeros(nr + 2), np.zeros(max(nv, 1))
    dim = 5 + 7 * nv
    cov = np.zeros((dim, dim)) if opts.core.compute_covariance else None
    lo = hi = None
    if bounds is not None:
        lo = np.array([bounds.fx_min, bounds.fy_min, bounds.cx_min, bounds.cy_min, bounds.skew_min], dtype=np.float64)
        hi = np.array([bounds.fx_max, bounds.fy_max, bounds.cx_max, bounds.cy_max, bounds.skew_max], dtype=np.float64)
    fi = np.ascontiguousarray(list(fixed_distortion_indices), dtype=np.int32)
    fv = np.ascontiguousarray(list(fixed_distortion_values) + [0.0] * (len(fi) - len(list(fixed_distortion_values))), dtype=np.float64)
    capi.check(lib, lib.cba_optimize_intrinsics_semidlt(nv, i64ptr(off), dptr(X), dptr(Y), dptr(u), dptr(v), dptr(K), dptr(poses), nr,
                                                        dptr(lo), dptr(hi), i32ptr(fi) if len(fi) else i32ptr(None),
                                                        dptr(fv) if len(fi) else dptr(None), len(fi), C.byref(copts), C.byref(s),
                                                        dptr(dist), dptr(ve), dptr(cov)))
    if nv < 4:  # the reference returns a default-constructed result
        return IntrinsicsOptimizationResult(OptimResult(), np.zeros(10), [], [])
    camera = np.concatenate([K, np.zeros(5)])
    camera[5:5 + nr] = dist[:nr]  # BrownConrady coeffs [k1..k_nr, p1, p2] laid into the 10-vector [.., k1 k2 k3 p1 p2]
    camera[8:10] = dist[nr:]
    c = cov if cov is not None and np.any(cov) else None
    return IntrinsicsOptimizationResult(result_core(s, c), camera, [pose_to_matrix(p) for p in poses], [float(e) for e in ve[:nv]], dist)


def optimize_intrinsics_semidlt_sharded(local_views, first_view: int, n_views_total: int, initial_guess, init_c_se3_t, n_ranks: int,
                                        rank: int, allreduce=None, rccl_id: Optional[bytes] = None,
                                        opts: Optional[IntrinsicsOptimOptions] = None, bounds: Optional[CalibrationBounds] = None,
                                        fixed_distortion_indices=(), fixed_distortion_values=(), device: int = 0
                                        ) -> IntrinsicsOptimizationResult:
    """optimize_intrinsics_semidlt with the VIEWS sharded over ranks (cba_optimize_intrinsics_semidlt_sharded / _rccl): this rank
    passes the observations of views [first_view, first_view + len(local_views)) and the seeds of ALL n_views_total views
    (init_c_se3_t, the same on every rank).  Transport: ``allreduce(np.ndarray)`` (sums in place across ranks) or ``rccl_id``
    (the 128 bytes of rccl_unique_id() from one rank).  The result covers the whole problem and is identical on every rank."""
    opts = opts or IntrinsicsOptimOptions()
    lib = capi.load_library()
    nl, nv = len(local_views), int(n_views_total)
    vs = [np.asarray(v, dtype=np.float64).reshape(-1, 4) for v in local_views]
    off = np.zeros(nl + 1, dtype=np.int64)
    np.cumsum([v.shape[0] for v in vs], out=off[1:])
    allv = np.concatenate(vs, axis=0) if nl else np.zeros((0, 4))
    X, Y, u, v = (np.ascontiguousarray(allv[:, k]) for k in range(4))
    K = np.ascontiguousarray(np.asarray(initial_guess, dtype=np.float64).reshape(5)).copy()
    if len(init_c_se3_t) != nv:
        raise capi.CbaError(capi.CBA_ERR_INVALID_ARGUMENT, "one seed pose per view of the whole problem")
    poses = np.ascontiguousarray(np.stack([pose_from_matrix(T) for T in init_c_se3_t])) if nv else np.zeros((1, 7))
    nr = int(getattr(opts, "num_radial", 2))
    copts = to_cba_options(opts.core, optimize_skew=opts.optimize_skew)
    s = CbaSummary()
    dist, ve = np.zeros(nr + 2), np.zeros(max(nv, 1))
    dim = 5 + 7 * nv
    cov = np.zeros((dim, dim)) if opts.core.compute_covariance else None
    lo = hi = None
    if bounds is not None:
        lo = np.array([bounds.fx_min, bounds.fy_min, bounds.cx_min, bounds.cy_min, bounds.skew_min], dtype=np.float64)
        hi = np.array([bounds.fx_max, bounds.fy_max, bounds.cx_max, bounds.cy_max, bounds.skew_max], dtype=np.float64)
    fi = np.ascontiguousarray(list(fixed_distortion_indices), dtype=np.int32)
    fv = np.ascontiguousarray(list(fixed_distortion_values) + [0.0] * (len(fi) - len(list(fixed_distortion_values))), dtype=np.float64)
    head = (nl, i64ptr(off), dptr(X), dptr(Y), dptr(u), dptr(v), nv, int(first_view), dptr(K), dptr(poses), nr, dptr(lo), dptr(hi),
            i32ptr(fi) if len(fi) else i32ptr(None), dptr(fv) if len(fi) else dptr(None), len(fi), C.byref(copts), C.byref(s), dptr(dist),
            dptr(ve), dptr(cov))
    if rccl_id is not None:
        idb = (C.c_uint8 * capi.RCCL_UNIQUE_ID_BYTES).from_buffer_copy(bytes(rccl_id))
        capi.check(lib, lib.cba_optimize_intrinsics_semidlt_rccl(*head, idb, int(n_ranks), int(rank), int(device)))
    else:
        def _cb(buf, count, _user):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(int(count),)))
                return 0
            except Exception:  # pragma: no cover
                return 1

        cb = capi.ALLREDUCE_FN(_cb)
        capi.check(lib, lib.cba_optimize_intrinsics_semidlt_sharded(*head, cb, None, int(n_ranks), int(rank), int(device)))
    if nv < 4:
        return IntrinsicsOptimizationResult(OptimResult(), np.zeros(10), [], [])
    camera = np.concatenate([K, np.zeros(5)])
    camera[5:5 + nr] = dist[:nr]
    camera[8:10] = dist[nr:]
    c = cov if cov is not None and np.any(cov) else None
    return IntrinsicsOptimizationResult(result_core(s, c), camera, [pose_to_matrix(p) for p in poses], [float(e) for e in ve[:nv]], dist)


def estimate_homography_batch(views):
    """Batched estimate_homography (DLT path, homography.cpp:31-43): -> (list of 3x3 H, list of success flags)."""
    lib = capi.load_library()
    nv = len(views)
    vs = [np.asarray(v, dtype=np.float64).reshape(-1, 4) for v in views]
    off = np.zeros(nv + 1, dtype=np.int64)
    np.cumsum([v.shape[0] for v in vs], out=off[1:])
    allv = np.concatenate(vs, axis=0) if nv else np.zeros((0, 4))
    X, Y, u, v = (np.ascontiguousarray(allv[:, k]) for k in range(4))
    H = np.zeros((max(nv, 1), 9))
    ok = np.zeros(max(nv, 1), dtype=np.int32)
    capi.check(lib, lib.cba_estimate_homography_batch(nv, i64ptr(off), dptr(X), dptr(Y), dptr(u), dptr(v), dptr(H), i32ptr(ok)))
    return [H[i].reshape(3, 3).copy() for i in range(nv)], [bool(k) for k in ok[:nv]]


def estimate_planar_pose_batch(views, intrinsics) -> List[np.ndarray]:
    """Batched estimate_planar_pose (linear/planarpose.h, planarpose_linear.cpp:54-76) on the GPU: one 4x4 c_T_t per view."""
    lib = capi.load_library()
    nv = len(views)
    vs = [np.asarray(v, dtype=np.float64).reshape(-1, 4) for v in views]
    off = np.zeros(nv + 1, dtype=np.int64)
    np.cumsum([v.shape[0] for v in vs], out=off[1:])
    allv = np.concatenate(vs, axis=0) if nv else np.zeros((0, 4))
    X, Y, u, v = (np.ascontiguousarray(allv[:, k]) for k in range(4))
    K = np.ascontiguousarray(np.asarray(intrinsics, dtype=np.float64).reshape(-1)[:5]).copy()
    poses = np.zeros((max(nv, 1), 7))
    capi.check(lib, lib.cba_estimate_planar_pose_batch(nv, i64ptr(off), dptr(X), dptr(Y), dptr(u), dptr(v), dptr(K), dptr(poses)))
    return [pose_to_matrix(p) for p in poses[:nv]]
