"""Mode B rate of several handles of the same problem created one after the other in ONE process (kept alive): does the placement of a
handle's buffers matter for the compute-bound kernel too?"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calibration_amd import synth, optim
which = sys.argv[1] if len(sys.argv) > 1 else "c3q"
sc = synth.scene_extrinsics_shard(4000, 0, 1000) if which == "c3q" else synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
alive = []
for k in range(6):
    h = optim.ReprojHandle(sc.flat); alive.append(h)
    print(f"{which} handle #{k}: Mode B {min(h.normal_eq_timed(2, 10) for _ in range(3)):.4f} ms", flush=True)
print("again: " + " ".join(f"{min(h.normal_eq_timed(2, 10) for _ in range(3)):.4f}" for h in alive), flush=True)
