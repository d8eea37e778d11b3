#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per k_eval launch from two rocprofv3 --pmc passes of tools/eval_only.py (one counter per pass, as
MI355X_MICROARCH.md prescribes), with the gfx950 correction (FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads).
usage: python tools/pmc_eval_summary.py <fetch dir> <write dir> <algorithmic bytes per launch> <label> > profiles/<name>.json"""
import csv, glob, json, os, sys


def per_launch(directory, counter):
    f = sorted(glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True))[-1]
    by_dispatch, names = {}, set()
    for r in csv.DictReader(open(f)):
        if "k_eval" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            by_dispatch[r["Dispatch_Id"]] = by_dispatch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            names.add(r["Kernel_Name"][:60])
    vals = list(by_dispatch.values())
    return vals, sorted(names)


fd, wd, alg, label = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # k_eval dispatches per evaluation pass (segments); 0 = one
fv, names = per_launch(fd, "FETCH_SIZE")
wv, _ = per_launch(wd, "WRITE_SIZE")
seg = max(1, launches)
fetch, write = sum(fv) / (len(fv) / seg), sum(wv) / (len(wv) / seg)
print(json.dumps({"workload": label, "kernel": names, "dispatches_per_pass": seg, "passes_seen": [len(fv) / seg, len(wv) / seg],
                  "FETCH_SIZE_KiB_per_pass": fetch, "WRITE_SIZE_KiB_per_pass": write, "read_bytes_per_pass": 2 * fetch * 1024,
                  "write_bytes_per_pass": write * 1024, "hbm_bytes_per_pass": 2 * fetch * 1024 + write * 1024,
                  "algorithmic_bytes_per_pass": alg, "traffic_over_algorithmic": (2 * fetch * 1024 + write * 1024) / alg,
                  "note": "FETCH_SIZE / WRITE_SIZE in KiB; gfx950: read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section)"}, indent=1))
