"""Timing ablation of Mode A for the two-pose chain (P = 22): full kernel / no arithmetic / no loads (see exp_abl.py for P = 16)."""
import os, sys, statistics
sys.path.insert(0, ".")
from calibration_amd import synth, optim
sc = synth.scene_extrinsics(200, 8, rows=50, cols=100, spacing=0.008, noise_px=0.2, seed=137)
h = optim.ReprojHandle(sc.flat)
n = sc.flat.n_obs
os.environ["CBA_EVAL_VARIANT"], os.environ["CBA_EVAL_BLOCKED"] = "1", "1"
res = {a: [] for a in (0, 1, 2)}
for rnd in range(8):
    for a in (0, 1, 2):
        os.environ["CBA_EVAL_ABLATE"] = str(a)
        res[a].append(h.eval_timed(2, 20))
for a, name in ((0, "full"), (1, "no arithmetic (loads + stores)"), (2, "no loads (arithmetic + stores)")):
    m = min(res[a]); print(f"{name:34s}: min {m:.4f} ms ({400 * n / m / 1e6:.0f} GB/s alg)  median {statistics.median(res[a]):.4f}")
