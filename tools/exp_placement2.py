"""k_eval rate of several handles of the same C2 problem created one after the other in ONE process: (a) all kept alive, (b) each
released before the next is created."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calibration_amd import synth, optim
sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
def rate(h):
    h.eval_timed(3, 5)
    return min(h.eval_timed(1, 20) for _ in range(8))
alive = []
for k in range(6):
    h = optim.ReprojHandle(sc.flat); alive.append(h)
    print(f"alive    #{k}: {rate(h):.4f} ms ({304e7 / rate(h) / 1e6:.0f} GB/s)", flush=True)
print("again, in creation order: " + " ".join(f"{304e7 / rate(h) / 1e6:.0f}" for h in alive), flush=True)
for h in alive: h.close()
for k in range(6):
    with optim.ReprojHandle(sc.flat) as h:
        print(f"released #{k}: {rate(h):.4f} ms ({304e7 / rate(h) / 1e6:.0f} GB/s)", flush=True)
