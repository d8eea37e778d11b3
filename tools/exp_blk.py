"""k_eval variants timed on the SAME buffers (knobs re-read per timed call), interleaved rounds."""
import os, sys, statistics
sys.path.insert(0, ".")
from calibration_amd import synth, optim
sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1, 1), (3, 1), (5, 1), (7, 1), (0, 1), (2, 1), (4, 1)]
h = optim.ReprojHandle(sc.flat)
res = {c: [] for c in cfgs}
for rnd in range(8):
    for c in cfgs:
        os.environ["CBA_EVAL_VARIANT"], os.environ["CBA_EVAL_BLOCKED"] = str(c[0]), str(c[1])
        res[c].append(h.eval_timed(2, 20))
for c in cfgs:
    m, md = min(res[c]), statistics.median(res[c])
    print(f"  variant {c[0]} blocked {c[1]}: min {m:.4f} ms ({304e7/m/1e6:.0f} GB/s)  median {md:.4f} ms ({304e7/md/1e6:.0f} GB/s)")
