#!/usr/bin/env python3
"""Per-phase times of the resident LM kernel (CBA_LM_RESIDENT_PROFILE=1, printed by the library on stderr) for one small problem
of each chain."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["CBA_LM_RESIDENT_PROFILE"] = "1"
os.environ["CBA_LM_RESIDENT"] = "2"
from calibration_amd import optim, synth, capi
o = capi.default_options(); o.compute_covariance = 0
for name, sc in (("intr 20 x 88", synth.scene_intrinsics(20, noise_px=0.2)), ("ext 10 x 2 x 88", synth.scene_extrinsics(10, 2, noise_px=0.2)),
                 ("bundle 25 x 88", synth.scene_bundle(25, 1, noise_px=0.2))):
    with optim.ReprojHandle(sc.flat) as h:
        s = h.solve(o)
    print(f"{name}: {s.iterations} iterations, {s.solve_seconds*1e3:.2f} ms", file=sys.stderr, flush=True)
