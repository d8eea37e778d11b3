"""Does the position of the Mode A output buffer in the device heap change the write rate?  One process per pre-allocation size:
`python tools/exp_placement.py <GiB held before the handle is created>` -> ms per k_eval launch at C2."""
import os, sys, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from calibration_amd import synth, optim
pre = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
hold = torch.empty(int(pre * (1 << 30)), dtype=torch.uint8, device="cuda") if pre > 0 else None
sc = synth.scene_intrinsics(1000, rows=100, cols=100, spacing=0.008, noise_px=0.2)
with optim.ReprojHandle(sc.flat) as h:
    h.eval_timed(3, 5)
    t = [h.eval_timed(1, 20) for _ in range(12)]
print(f"pre-allocated {pre:5.1f} GiB: min {min(t):.4f} ms ({304e7 / min(t) / 1e6:.0f} GB/s)  median {statistics.median(t):.4f} ms", flush=True)
