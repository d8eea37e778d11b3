"""ctypes binding of ``include/calibba.h`` (libcalibba.so).

This is the only place the shared library is loaded.  Loading fails loudly when the library has
not been built (``python -c 'import __graft_entry__ as g; g.build()'``); compute calls fail loudly
(``CbaError`` with ``CBA_ERR_NO_DEVICE``) when no GPU is visible.  Nothing here falls back to a CPU
implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

CBA_OK, CBA_ERR_INVALID_ARGUMENT, CBA_ERR_RUNTIME, CBA_ERR_NO_DEVICE, CBA_ERR_HIP, CBA_ERR_INTERNAL = range(6)
CHAIN_INTRINSIC, CHAIN_EXTRINSIC, CHAIN_BUNDLE = 0, 1, 2
CAMERA_PINHOLE_BC, CAMERA_SCHEIMPFLUG = 0, 1
TERM_CONVERGENCE, TERM_NO_CONVERGENCE, TERM_FAILURE = 0, 1, 2
RCCL_UNIQUE_ID_BYTES = 128

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class CbaOptions(C.Structure):
    """``cba_options`` — OptimOptions (optimize.h:24-33) + per-stage switches."""

    _fields_ = [
        ("optimizer", C.c_int32),
        ("max_iterations", C.c_int32),
        ("huber_delta", C.c_double),
        ("epsilon", C.c_double),
        ("compute_covariance", C.c_int32),
        ("verbose", C.c_int32),
        ("optimize_intrinsics", C.c_int32),
        ("optimize_skew", C.c_int32),
        ("optimize_extrinsics", C.c_int32),
        ("optimize_target_pose", C.c_int32),
    ]


class CbaSummary(C.Structure):
    """``cba_summary`` — OptimResult (optimize.h:35-40) without the covariance matrix."""

    _fields_ = [
        ("success", C.c_int32),
        ("termination", C.c_int32),
        ("iterations", C.c_int32),
        ("successful_steps", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("solve_seconds", C.c_double),
        ("report", C.c_char * 192),
    ]


class CbaReprojProblem(C.Structure):
    """``cba_reproj_problem`` — one reprojection bundle problem in SoA form."""

    _fields_ = [
        ("chain", C.c_int32),
        ("camera_model", C.c_int32),
        ("n_blocks", C.c_int32),
        ("n_cams", C.c_int32),
        ("n_views", C.c_int32),
        ("reserved0", C.c_int32),
        ("first_view_global", C.c_int64),
        ("blk_offset", c_int64_p),
        ("blk_cam", c_int32_p),
        ("blk_view", c_int32_p),
        ("blk_b_T_g", c_double_p),
        ("X", c_double_p),
        ("Y", c_double_p),
        ("u", c_double_p),
        ("v", c_double_p),
        ("intr", c_double_p),
        ("cam_pose", c_double_p),
        ("view_pose", c_double_p),
        ("target_pose", c_double_p),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, c_double_p, C.c_int64, C.c_void_p)


class CbaError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libcalibba status {status}: {message}")
        self.status = status
        self.message = message


class CbaInvalidArgument(CbaError, ValueError):
    """Maps the reference's std::invalid_argument."""


def dptr(a: Optional[np.ndarray]):
    if a is None:
        return C.cast(None, c_double_p)
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def i32ptr(a: Optional[np.ndarray]):
    if a is None:
        return C.cast(None, c_int32_p)
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_int32_p)


def i64ptr(a: Optional[np.ndarray]):
    if a is None:
        return C.cast(None, c_int64_p)
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_int64_p)


def library_path() -> str:
    return os.environ.get("CALIBBA_LIBRARY", os.path.join(_HERE, "lib", "libcalibba.so"))


_LIB = None

# name -> (restype, argtypes); every symbol include/calibba.h declares
PROTOTYPES = {
    "cba_version": (C.c_char_p, []),
    "cba_last_error": (C.c_char_p, []),
    "cba_device_count": (C.c_int32, []),
    "cba_trim_cache": (None, []),
    "cba_set_device": (C.c_int32, [C.c_int32]),
    "cba_get_device": (C.c_int32, []),
    "cba_options_default": (None, [C.POINTER(CbaOptions)]),
    "cba_intrinsics_size": (C.c_int32, [C.c_int32]),
    "cba_local_columns": (C.c_int32, [C.c_int32, C.c_int32]),
    "cba_pose_from_matrix": (None, [c_double_p, c_double_p]),
    "cba_pose_to_matrix": (None, [c_double_p, c_double_p]),
    "cba_reproj_create": (C.c_int32, [C.POINTER(CbaReprojProblem), C.c_int32, C.POINTER(C.c_void_p)]),
    "cba_reproj_create_aos": (C.c_int32, [C.POINTER(CbaReprojProblem), C.POINTER(c_double_p), C.c_int32, C.POINTER(C.c_void_p)]),
    "cba_reproj_destroy": (None, [C.c_void_p]),
    "cba_reproj_set_params": (C.c_int32, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "cba_reproj_get_params": (C.c_int32, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "cba_reproj_num_observations": (C.c_int64, [C.c_void_p]),
    "cba_reproj_eval": (C.c_int32, [C.c_void_p]),
    "cba_reproj_eval_fetch": (C.c_int32, [C.c_void_p, c_double_p, c_double_p]),
    "cba_reproj_eval_fetch_blocks": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, c_double_p, c_double_p]),
    "cba_reproj_eval_timed": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, c_double_p]),
    "cba_reproj_normal_eq_timed": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, c_double_p]),
    "cba_reproj_set_scalar": (C.c_int32, [C.c_void_p, C.c_int32]),
    "cba_reproj_eval_fetch_f32": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "cba_reproj_cost": (C.c_int32, [C.c_void_p, C.c_double, c_double_p]),
    "cba_reproj_block_normal_eq": (C.c_int32, [C.c_void_p, c_double_p]),
    "cba_reproj_block_normal_eq_size": (C.c_int64, [C.c_void_p]),
    "cba_reproj_solve": (C.c_int32, [C.c_void_p, C.POINTER(CbaOptions), C.POINTER(CbaSummary)]),
    "cba_reproj_set_lm_mode": (C.c_int32, [C.c_void_p, C.c_int32]),
    "cba_reproj_solve_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)]),
    "cba_reproj_covariance_dim": (C.c_int64, [C.c_void_p]),
    "cba_reproj_covariance": (C.c_int32, [C.c_void_p, C.POINTER(CbaOptions), c_double_p]),
    "cba_reproj_covariance_shared_dim": (C.c_int64, [C.c_void_p]),
    "cba_reproj_covariance_shared": (C.c_int32, [C.c_void_p, C.POINTER(CbaOptions), c_double_p]),
    "cba_reproj_covariance_views": (C.c_int32, [C.c_void_p, C.POINTER(CbaOptions), C.c_int32, c_int32_p, c_double_p]),
    "cba_reproj_set_allreduce": (C.c_int32, [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int32, C.c_int32]),
    "cba_rccl_unique_id": (C.c_int32, [C.POINTER(C.c_uint8)]),
    "cba_reproj_init_rccl": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]),
    "cba_optimize_intrinsics": (
        C.c_int32,
        [C.c_int32, C.c_int32, c_int64_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
         C.POINTER(CbaOptions), C.POINTER(CbaSummary), c_double_p],
    ),
    "cba_optimize_extrinsics": (
        C.c_int32,
        [C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_int64_p, c_int32_p, c_int32_p, c_double_p, c_double_p,
         c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(CbaOptions), C.POINTER(CbaSummary),
         c_double_p],
    ),
    "cba_optimize_bundle": (
        C.c_int32,
        [C.c_int32, C.c_int32, C.c_int32, c_int64_p, c_int32_p, c_double_p, c_double_p, c_double_p, c_double_p,
         c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(CbaOptions), C.POINTER(CbaSummary), c_double_p],
    ),
    "cba_optimize_planar_pose": (
        C.c_int32,
        [C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int32, c_double_p, C.POINTER(CbaOptions),
         C.POINTER(CbaSummary), c_double_p, c_double_p, c_double_p],
    ),
    "cba_optimize_planar_pose_batch": (
        C.c_int32,
        [C.c_int32, c_int64_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int32, c_double_p,
         C.POINTER(CbaOptions), C.POINTER(CbaSummary), c_double_p, c_double_p, c_double_p],
    ),
    "cba_optimize_homography": (
        C.c_int32,
        [C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(CbaOptions), C.POINTER(CbaSummary),
         c_double_p],
    ),
    "cba_optimize_homography_batch": (
        C.c_int32,
        [C.c_int32, c_int64_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(CbaOptions),
         C.POINTER(CbaSummary), c_double_p],
    ),
    "cba_optimize_intrinsics_semidlt": (
        C.c_int32,
        [C.c_int32, c_int64_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int32, c_double_p, c_double_p,
         c_int32_p, c_double_p, C.c_int32, C.POINTER(CbaOptions), C.POINTER(CbaSummary), c_double_p, c_double_p, c_double_p],
    ),
    "cba_optimize_intrinsics_semidlt_sharded": (
        C.c_int32,
        [C.c_int32, c_int64_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int32, C.c_int32, c_double_p, c_double_p, C.c_int32,
         c_double_p, c_double_p, c_int32_p, c_double_p, C.c_int32, C.POINTER(CbaOptions), C.POINTER(CbaSummary), c_double_p, c_double_p,
         c_double_p, ALLREDUCE_FN, C.c_void_p, C.c_int32, C.c_int32, C.c_int32],
    ),
    "cba_optimize_intrinsics_semidlt_rccl": (
        C.c_int32,
        [C.c_int32, c_int64_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int32, C.c_int32, c_double_p, c_double_p, C.c_int32,
         c_double_p, c_double_p, c_int32_p, c_double_p, C.c_int32, C.POINTER(CbaOptions), C.POINTER(CbaSummary), c_double_p, c_double_p,
         c_double_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32],
    ),
    "cba_estimate_homography_batch": (
        C.c_int32, [C.c_int32, c_int64_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_int32_p]),
    "cba_estimate_planar_pose_batch": (
        C.c_int32, [C.c_int32, c_int64_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "cba_estimate_handeye_dlt": (C.c_int32, [C.c_int32, c_double_p, c_double_p, C.c_double, c_double_p]),
    "cba_estimate_and_optimize_handeye": (
        C.c_int32,
        [C.c_int32, c_double_p, c_double_p, C.c_double, c_double_p, C.POINTER(CbaOptions), C.POINTER(CbaSummary), c_double_p],
    ),
    "cba_estimate_and_optimize_handeye_sharded": (
        C.c_int32, [C.c_int32, c_double_p, c_double_p, C.c_double, C.c_int32, c_double_p, C.POINTER(CbaOptions), C.POINTER(CbaSummary),
                    c_double_p, ALLREDUCE_FN, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "cba_estimate_and_optimize_handeye_rccl": (
        C.c_int32, [C.c_int32, c_double_p, c_double_p, C.c_double, C.c_int32, c_double_p, C.POINTER(CbaOptions), C.POINTER(CbaSummary),
                    c_double_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32]),
    "cba_optimize_handeye": (
        C.c_int32,
        [C.c_int32, c_double_p, c_double_p, c_double_p, C.POINTER(CbaOptions), C.POINTER(CbaSummary), c_double_p],
    ),
}


def bind(lib, prototypes=None):
    for name, (res, args) in (prototypes or PROTOTYPES).items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


def load_library(path: Optional[str] = None):
    """Load libcalibba.so and bind every symbol of calibba.h.  Raises if it is not built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or library_path()
    # PyTorch-ROCm wheels bundle their own libamdhip64 / librccl / libhsa-runtime64 (same SONAMEs as the
    # system ROCm libraries libcalibba.so links).  If libcalibba is loaded first and torch later, glibc maps
    # BOTH copies (torch asks for the unversioned file names) and the process ends up with two HIP runtimes.
    # Importing torch first makes libcalibba bind to the copies torch already mapped: one HIP runtime, and
    # the same RCCL that torch.distributed's "nccl" backend uses.
    try:
        import torch  # noqa: F401
    except ImportError:  # a host without PyTorch: the system ROCm libraries are used
        pass
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
            "calibration_amd has no CPU fallback."
        )
    lib = bind(C.CDLL(p, mode=C.RTLD_GLOBAL))
    if path is None:
        _LIB = lib
    return lib


def check(lib, status: int):
    if status == CBA_OK:
        return
    msg = lib.cba_last_error()
    msg = msg.decode("utf-8", "replace") if msg else ""
    if status == CBA_ERR_INVALID_ARGUMENT:
        raise CbaInvalidArgument(status, msg)
    raise CbaError(status, msg)


def default_options(lib=None) -> CbaOptions:
    o = CbaOptions()
    (lib or load_library()).cba_options_default(C.byref(o))
    return o
