"""Interleaved A/B timing of k_eval variants in ONE process (one handle per variant, variant chosen by
CBA_EVAL_VARIANT at handle creation), N rounds; prints min / median ms per eval."""
import os, sys, statistics
sys.path.insert(0, ".")
from calibration_amd import synth, optim
variants = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]
shape = os.environ.get("EXP_SHAPE", "c2")  # c2 (P = 16), c5 (Scheimpflug, P = 18), c3q (8-camera rig / 4, P = 22)
if shape == "c5":
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2, model=1, seed=5)
elif shape == "c3q":
    sc = synth.scene_extrinsics_shard(4000, 0, 1000)
else:
    sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
bytes_per = {"c2": 304, "c5": 336, "c3q": 400}[shape] * sc.flat.n_obs
hs = {}
for v in variants:
    os.environ["CBA_EVAL_VARIANT"] = str(v)
    hs[v] = optim.ReprojHandle(sc.flat)
    hs[v].eval_timed(3, 5)
res = {v: [] for v in variants}
for rnd in range(12):
    for v in variants:
        res[v].append(hs[v].eval_timed(1, 20))
for v in variants:
    m, md = min(res[v]), statistics.median(res[v])
    print(f"{shape} variant {v}: min {m:.4f} ms ({bytes_per/m/1e6:.0f} GB/s)  median {md:.4f} ms ({bytes_per/md/1e6:.0f} GB/s)")
