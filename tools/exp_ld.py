"""Column-stride experiment for k_eval: (ld*8) mod 2 MiB forced to given byte offsets (CBA_LD_MOD),
variant 5 (nt, 4 tiles/wave) and 0; interleaved rounds in one process."""
import os, sys, statistics
sys.path.insert(0, ".")
from calibration_amd import synth, optim
mods = [int(a) for a in sys.argv[1:]] or [-1, 0, 256, 512, 2048, 4096, 4352, 65792, 131328]
sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
res = {}
for variant in (0, 5):
    os.environ["CBA_EVAL_VARIANT"] = str(variant)
    for m in mods:
        if m < 0: os.environ.pop("CBA_LD_MOD", None)
        else: os.environ["CBA_LD_MOD"] = str(m)
        with optim.ReprojHandle(sc.flat) as h:
            h.eval_timed(3, 5)
            t = [h.eval_timed(1, 20) for _ in range(5)]
        res[(variant, m)] = min(t)
        print(f"variant {variant} ld_mod {m:7d}: min {min(t):.4f} ms ({304e7/min(t)/1e6:.0f} GB/s) median {statistics.median(t):.4f}", flush=True)
