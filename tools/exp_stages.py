#!/usr/bin/env python3
"""Where a C1-sized calibration call spends its time: handle creation / solve / covariance / destruction, through the handle API."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import copy
import numpy as np
from calibration_amd import optim, synth, capi
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
