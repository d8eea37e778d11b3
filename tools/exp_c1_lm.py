#!/usr/bin/env python3
"""LM wall time of a C1-sized problem (20 views x 88 points) solved three times on one handle: with CBA_LM_GRAPH=1 the first
solve pays the HIP-graph capture + instantiation of the three stages, the later ones replay them."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calibration_amd import optim, synth, capi
sc = synth.scene_intrinsics(20, noise_px=0.2)
init = (sc.flat.intr.copy(), sc.flat.view_pose.copy())
o = capi.default_options(); o.compute_covariance = 0
with optim.ReprojHandle(sc.flat) as h:
    for k in range(3):
        h.set_params(intr=init[0], view_pose=init[1])
        t0 = time.perf_counter(); s = h.solve(o); dt = time.perf_counter() - t0
        print(f"CBA_LM_GRAPH={os.environ.get('CBA_LM_GRAPH', 'default')} CBA_LM_RESIDENT={os.environ.get('CBA_LM_RESIDENT', 'default')} solve {k}: "
              f"{dt*1e3:.2f} ms ({s.solve_seconds*1e3:.2f} in the engine), {s.iterations} iterations, {dt/s.iterations*1e6:.0f} us/iteration, "
              f"cost {s.final_cost:.9e}, {bytes(s.report).split(b':')[0].decode()}")
