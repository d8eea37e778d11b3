#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of tools/profile_gpu.sh (FETCH_SIZE, WRITE_SIZE; one counter per pass, as
MI355X_MICROARCH.md prescribes) into profiles/pmc_k_eval.json: HBM bytes per k_eval launch with the gfx950 correction
(FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> read bytes = 2 * FETCH_SIZE KiB * 1024)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(directory, counter):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", directory, "*", "*counter_collection.csv")))[-1]
    by_dispatch = {}
    for r in csv.DictReader(open(f)):
        if "k_eval" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            by_dispatch[r["Dispatch_Id"]] = by_dispatch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    vals = list(by_dispatch.values())
    return sum(vals) / len(vals), len(vals)


fetch, nf = per_launch("prof_fetch", "FETCH_SIZE")
write, nw = per_launch("prof_write", "WRITE_SIZE")
out = {
    "workload": "pinhole+Brown-Conrady intrinsics, 1000 views x 10000 pts, fp64 (bench.py default)",
    "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-lm (one counter per pass)",
    "kernel": "cba::k_eval<0, 0, true, 1, true, 0, double>",
    "note": "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, so read "
            "bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section). X, Y are deduplicated across views (one 160 KB copy, "
            "cache-resident), so the HBM read side is u, v only.",
    "FETCH_SIZE_KiB_per_launch": fetch, "FETCH_SIZE_launches": nf,
    "WRITE_SIZE_KiB_per_launch": write, "WRITE_SIZE_launches": nw,
    "read_bytes_per_launch": 2 * fetch * 1024, "write_bytes_per_launch": write * 1024,
    "hbm_bytes_per_launch": 2 * fetch * 1024 + write * 1024, "algorithmic_bytes_per_launch": 304 * 10_000_000,
}
for name in ("pmc_k_eval.json", sys.argv[1] if len(sys.argv) > 1 else "r03_pmc_k_eval.json"):
    json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(json.dumps(out))
