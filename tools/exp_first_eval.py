import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from calibration_amd import synth, optim
sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
with optim.ReprojHandle(sc.flat) as h:
    t0 = time.perf_counter(); h.eval(); t1 = time.perf_counter(); h.eval(); t2 = time.perf_counter()
    print(f"CBA_EVAL_CONTIGUOUS={os.environ.get('CBA_EVAL_CONTIGUOUS','default (1)')}: first eval {1e3*(t1-t0):.1f} ms, second {1e3*(t2-t1):.2f} ms, rate {304e7/h.eval_timed(2,20)/1e6:.0f} GB/s")
