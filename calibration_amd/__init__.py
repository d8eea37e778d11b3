"""calibration_amd — MI355X-native bundle-adjustment engine (libcalibba) and its host-side mirror
of VitalyVorobyev/calibration's ``calib::estimation_optim`` refinement API.

The compute path is the HIP library behind ``include/calibba.h``; this package only loads it
(ctypes) and flattens host containers into the SoA buffers the C ABI takes.  There is no CPU
fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""
from .capi import (  # noqa: F401
    CbaError,
    CbaOptions,
    CbaReprojProblem,
    CbaSummary,
    load_library,
    library_path,
)
from .optim import (  # noqa: F401
    BundleObservation,
    BundleOptions,
    ExtrinsicOptions,
    IntrinsicsOptimOptions,
    OptimOptions,
    ReprojHandle,
    optimize_bundle,
    optimize_extrinsics,
    optimize_handeye,
    optimize_intrinsics,
    optimize_planar_pose,
    optimize_planar_pose_batch,
    PlanarPoseOptions,
)

__version__ = "0.1.0"
