#!/usr/bin/env python3
"""Turns the two VALU / LDS / wait counter passes of tools/profile_r0N.sh (gpurun_out/<tag>_prof_valu, <tag>_prof_valu2: separate
`rocprofv3 --pmc` runs of `bench.py --steps 5 --warmup 1 --no-cpu`, no tracing options) into
profiles/<tag>_pmc_valu_mode_a_b.json (usage: python tools/pmc_valu_summary.py [tag, default r03]): per-dispatch averages of the shared-rows Mode B kernel and of k_eval, the share of the launch's
cycles in which the vector pipe is active, and how a wavefront's cycles split.  Start / end timestamps of the dispatches give the
launch time (and with GRBM_GUI_ACTIVE the clock the chip held)."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"


def load(directory):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", directory, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    out = {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        big = int(r["Grid_Size"]) > 100000
        key = "k_ne_shared<DirectForm>" if ("k_ne_shared" in name and "DirectForm" in name and big) else \
              "k_ne_shared<MomentForm>" if ("k_ne_shared" in name and "MomentForm" in name and big) else ("k_eval" if "k_eval" in name else None)
        if key is None:
            continue
        d = out.setdefault(key, {}).setdefault(r["Dispatch_Id"], {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


p1, p2 = load(f"{TAG}_prof_valu"), load(f"{TAG}_prof_valu2")
res = {}
for k in ("k_ne_shared<DirectForm>", "k_ne_shared<MomentForm>", "k_eval"):
    if k not in p1 or k not in p2:
        continue
    a, b = list(p1[k].values()), list(p2[k].values())
    n = min(len(a), len(b))
    a, b = a[:n], b[:n]
    # the Mode B timing loop is 2 + 10 launches right after Mode A, then 200 + 50 in the settled clock state: split them
    groups = {"all": (a, b)}
    if "DirectForm" in k and len(a) > 100:
        groups = {"first 12 launches (right after the Mode A section)": (a[:12], b[:12]), "last 50 launches (clock settled)": (a[-50:], b[-50:])}
    for gname, (ga, gb) in groups.items():
        avg = lambda rows, c: sum(r[c] for r in rows) / len(rows)
        e = {c: avg(ga, c) for c in ("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES")}
        e.update({c: avg(gb, c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")})
        e["dispatches"] = len(ga)
        e["launch_us_under_pmc"] = avg(ga, "ns") / 1e3
        e["launch_cycles_per_xcd"] = e["GRBM_GUI_ACTIVE"] / 8
        e["clock_GHz"] = e["launch_cycles_per_xcd"] / (e["launch_us_under_pmc"] * 1e3)
        e["valu_active_fraction"] = 4 * e["SQ_ACTIVE_INST_VALU"] / (1024 * e["launch_cycles_per_xcd"])
        w = e["SQ_WAVE_CYCLES"]
        e["wave_time_split"] = {"valu": e["SQ_ACTIVE_INST_VALU"] / w, "lds": e["SQ_ACTIVE_INST_LDS"] / w, "issue_stall": e["SQ_WAIT_INST_ANY"] / w, "waiting": e["SQ_WAIT_ANY"] / w}
        res[k if gname == "all" else f"{k}, {gname}"] = e
out = {"command": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu (two passes: "
                  "{SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE}, {SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS "
                  "SQ_WAIT_INST_ANY SQ_WAIT_ANY}); no tracing options",
       "notes": "per-dispatch averages, summed over all XCDs / SEs / SIMDs.  SQ_ACTIVE_INST_*, SQ_WAVE_CYCLES, SQ_WAIT_* count quad-cycles.  VALU "
                "activity = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * launch cycles), launch cycles = GRBM_GUI_ACTIVE / 8 XCDs; clock = launch cycles / "
                "(End - Start timestamp of the dispatch).",
       "per_dispatch_average": res}
for d in ("profiles", "gpurun_out"):  # (gpurun brings back gpurun_out/ only)
    json.dump(out, open(os.path.join(ROOT, d, f"{TAG}_pmc_valu_mode_a_b.json"), "w"), indent=1)
for k, e in res.items():
    print(f"{k}: {e['dispatches']} dispatches, {e['launch_us_under_pmc']:.1f} us, {e['launch_cycles_per_xcd']:.0f} cycles -> {e['clock_GHz']:.2f} GHz, "
          f"VALU active {e['valu_active_fraction']:.3f}, INSTS_VALU {e['SQ_INSTS_VALU']:.3e}, split {', '.join(f'{a} {b:.2f}' for a, b in e['wave_time_split'].items())}")
