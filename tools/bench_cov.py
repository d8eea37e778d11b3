#!/usr/bin/env python3
"""Wall time of the shared-block covariance (cba_reproj_covariance_shared) vs the reference-layout full matrix at growing
view counts.  One JSON line per size."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calibration_amd import capi, optim
from tests import synth

for n_views, grid, full in ((100, 30, True), (1000, 30, True), (1000, 100, False), (4000, 30, False)):
    sc = synth.scene_intrinsics(n_views, rows=grid, cols=grid, spacing=0.8 / grid, noise_px=0.2)
    o = capi.default_options()
    with optim.ReprojHandle(sc.flat) as h:
        o.compute_covariance = 0
        h.solve(o)
        t0 = time.perf_counter(); cs = h.covariance_shared(o); t_sh = time.perf_counter() - t0
        rec = {"views": n_views, "points_per_view": grid * grid, "shared_dim": int(cs.shape[0]), "shared_s": t_sh,
               "sigma_fx_px": float(np.sqrt(cs[0, 0]))}
        if full:
            t0 = time.perf_counter(); cf = h.covariance(o); rec.update(full_dim=int(cf.shape[0]), full_s=time.perf_counter() - t0,
                                                                        leading_block_equal=bool(np.array_equal(cs, cf[:cs.shape[0], :cs.shape[0]])))
    print(json.dumps(rec), flush=True)
