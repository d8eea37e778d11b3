import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from calibration_amd import synth, optim, capi
from tests import helpers
lib = capi.load_library()
orc = helpers.load_oracle()
for mk, okw in ((lambda: synth.scene_extrinsics(4, 2, noise_px=0.2), {}),):
    a, b = mk(), mk()
    o = helpers.options(**okw)
    helpers.oracle_solve(orc, a.flat, o)
    with optim.ReprojHandle(b.flat) as h:
        s = h.solve(o)
        print(s.report)
        dim = int(lib.cba_reproj_covariance_dim(h.h)); cov = np.zeros((dim, dim))
        st = lib.cba_reproj_covariance(h.h, C.byref(o), capi.dptr(cov))
        print('status', st, lib.cba_last_error())
        print('param diff', helpers.param_diff(a.flat, b.flat))
