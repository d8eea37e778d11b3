import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from calibration_amd import synth, optim
sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
with optim.ReprojHandle(sc.flat) as h:
    for w, n in ((2, 10), (2, 10), (50, 50), (200, 50), (2, 10), (500, 100), (2, 10)):
        print(f"warmup {w:4d} iters {n:4d}: {h.normal_eq_timed(w, n):.4f} ms", flush=True)
    h.eval_timed(3, 50)
    print(f"after 53 Mode A launches, warmup 2 iters 10: {h.normal_eq_timed(2, 10):.4f} ms")
    print(f"again: {h.normal_eq_timed(2, 10):.4f} ms")
