"""Cost of the physically contiguous Mode A output block at larger sizes: first / second evaluation time and rate of the 8-camera rig with
<views> views (output = views x 8 x 5000 x 368 B)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calibration_amd import synth, optim
views = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
sc = synth.scene_extrinsics_shard(4000, 0, views)
with optim.ReprojHandle(sc.flat) as h:
    t0 = time.perf_counter(); h.eval(); t1 = time.perf_counter(); h.eval(); t2 = time.perf_counter()
    n = sc.flat.n_obs
    print(f"{views} views ({n * 368 / 2**30:.1f} GiB out) CBA_EVAL_CONTIGUOUS={os.environ.get('CBA_EVAL_CONTIGUOUS', 'default (1)')}: first eval "
          f"{1e3 * (t1 - t0):.1f} ms, second {1e3 * (t2 - t1):.2f} ms, rate {400 * n / h.eval_timed(1, 5) / 1e6:.0f} GB/s", flush=True)
