"""Random far-start problems over 2-4 in-process ranks (threads; barrier + fixed-order sum as the all-reduce): every rank must issue
the same exchange sequence, end bit-identical in the replicated blocks and reproduce the 1-rank solve.  Backend: the GPU engine
(default) or the test-only host-math build of the same LM driver (--host, runs without a GPU).
usage: python tools/fuzz_multirank.py [--host] [--cases N] [--first K]  ->  one JSON summary line"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from calibration_amd import capi, optim  # noqa: E402
from tests import synth  # noqa: E402
from calibration_amd.capi import CbaSummary  # noqa: E402
from tests import helpers  # noqa: E402
from tests import test_multirank_gloo as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--host", action="store_true")
ap.add_argument("--cases", type=int, default=300)
ap.add_argument("--first", type=int, default=0)
args = ap.parse_args()
hm = helpers.load_hostmath() if args.host else None
tot = dict(cases=0, exchanges=0, iterations=0, speculative=0, hits=0, misses=0, rejected=0, line_searches=0, line_search_evaluations=0,
           replicated_blocks_differ=0, termination_differs=0, cost_differs=0, params_differ=0, count_identity_broken=0, protocol_errors=0)
bad = []
for case in range(args.first, args.first + args.cases):
    kind, world, okw, make = T._random_case(case)
    ref = make().flat
    full = make().flat
    flats = [synth.shard_views(full, r, world) for r in range(world)]
    o = helpers.options(epsilon=1e-10, **okw)
    try:
        if args.host:
            d = ref.struct()
            s1 = CbaSummary()
            assert hm.hm_reproj_solve(C.byref(d), C.byref(o), capi.ALLREDUCE_FN(), None, 1, 0, C.byref(s1)) == 0
            res = T._threaded_solve(hm, flats, okw)
            res = [(s, dict(zip(("allreduce_calls", "bytes", "speculative_steps", "speculation_hits", "speculation_misses", "rejected_steps",
                                 "line_searches", "line_search_evaluations"), xs))) for s, xs in res]
        else:
            with optim.ReprojHandle(ref, device=0) as h:
                s1 = h.solve(o)

            def body(rank, reduce):
                with optim.ReprojHandle(flats[rank], device=0) as h:
                    h.set_allreduce(reduce, world, rank)
                    s = h.solve(helpers.options(epsilon=1e-10, **okw))
                    return s, h.solve_stats()

            res = T._in_process_ranks(world, body, timeout=120.0)
    except AssertionError as ex:
        tot["protocol_errors"] += 1
        bad.append(dict(case=case, error=str(ex)[:200]))
        continue
    s, xs = res[0]
    tot["cases"] += 1
    tot["exchanges"] += xs["allreduce_calls"]
    tot["iterations"] += s.iterations
    tot["speculative"] += xs["speculative_steps"]
    tot["hits"] += xs["speculation_hits"]
    tot["misses"] += xs["speculation_misses"]
    tot["rejected"] += xs["rejected_steps"]
    tot["line_searches"] += xs["line_searches"]
    tot["line_search_evaluations"] += xs["line_search_evaluations"]
    flags = []
    for (sr, xr), f in zip(res, flats):
        if not np.array_equal(f.intr, flats[0].intr) or (f.cam_pose is not None and not np.array_equal(f.cam_pose, flats[0].cam_pose)):
            flags.append("replicated_blocks_differ")
        if sr.termination != s1.termination:
            flags.append("termination_differs")
        if abs(sr.final_cost - s1.final_cost) > 1e-8 * max(1.0, s1.final_cost):
            flags.append("cost_differs")
        if helpers.rel_diff(ref.intr, f.intr) > 1e-6:
            flags.append("params_differ")
        if sr.iterations and xr["allreduce_calls"] != 1 + sr.iterations + xr["speculation_misses"] + xr["rejected_steps"] + xr[
                "line_search_evaluations"] + (sr.successful_steps - xr["speculation_hits"] - xr["speculation_misses"]):
            flags.append("count_identity_broken")
    for fl in set(flags):
        tot[fl] += 1
    if flags:
        bad.append(dict(case=case, kind=kind, world=world, flags=sorted(set(flags)), cost_1rank=s1.final_cost, cost=s.final_cost,
                        iters_1rank=s1.iterations, iters=s.iterations))
print(json.dumps(dict(backend="host-math" if args.host else "gpu", first=args.first, **tot, flagged=bad)))
