import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from calibration_amd import optim, synth
for model in (0, 1):
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2, model=model, seed=5)
    with optim.ReprojHandle(sc.flat) as h:
        a = min(h.normal_eq_timed(2, 10) for _ in range(3))
        h.set_scalar(1)
        b = min(h.normal_eq_timed(2, 10) for _ in range(3))
    print(f"model {model}: Mode B fp64 {a:.3f} ms, fp32 rows {b:.3f} ms")
