"""Phase timing of cba_reproj_create at C3 size (CBA_CREATE_TIMING=1 prints the phases to stderr)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CBA_CREATE_TIMING"] = "1"
from calibration_amd import optim, synth
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sc = synth.scene_extrinsics(int(4000 * scale), 8, rows=50, cols=100, spacing=0.008, noise_px=0.2, seed=137)
for k in range(2):
    t0 = time.perf_counter(); h = optim.ReprojHandle(sc.flat); print(f"create {k}: {time.perf_counter() - t0:.3f} s for {sc.flat.n_obs} observations", flush=True); h.close()
# the same from {X, Y, u, v} records read in place (cba_reproj_create_aos): what a binding to the reference's PlanarView pays
import numpy as np
f = sc.flat
t0 = time.perf_counter()
rec = np.empty((f.n_obs, 4)); rec[:, 0] = f.X; rec[:, 1] = f.Y; rec[:, 2] = f.u; rec[:, 3] = f.v
records = [rec[a:b] for a, b in zip(f.blk_offset[:-1], f.blk_offset[1:])]
print(f"(building the record arrays for this experiment: {time.perf_counter() - t0:.3f} s — the copy a caller with AoS data avoids the reverse of)", flush=True)
for k in range(2):
    t0 = time.perf_counter(); h = optim.ReprojHandle(f, records=records); print(f"create from records {k}: {time.perf_counter() - t0:.3f} s", flush=True); h.close()
