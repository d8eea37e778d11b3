#!/usr/bin/env python3
"""One LM step's kernel timeline from a rocprofv3 --kernel-trace CSV: for every kernel between two launches of the controller
(k_lm_ctl) of the LAST solve in the trace, its duration and the idle gap before it (us).
usage: python tools/timeline.py <..._kernel_trace.csv> [step-index-from-the-end, default 3]"""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n)).replace("cba::", "")[:60]
ctl = [i for i, r in enumerate(rows) if "k_lm_ctl" in r["Kernel_Name"]]
if len(ctl) < 3:
    raise SystemExit("no controller launches in the trace")
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = ctl[-k - 1], ctl[-k]
prev_end = int(rows[a]["End_Timestamp"])
tot_k = tot_g = 0.0
print(f"{'kernel':62s} {'gap us':>8s} {'dur us':>8s}")
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap, dur = (s - prev_end) / 1e3, (e - s) / 1e3
    tot_k += dur
    tot_g += max(gap, 0.0)
    print(f"{short(r['Kernel_Name']):62s} {gap:8.1f} {dur:8.1f}")
    prev_end = e
print(f"step: kernels {tot_k:.1f} us, gaps {tot_g:.1f} us, total {tot_k + tot_g:.1f} us")
