"""bench.py's one-JSON-line contract, on one GPU and as a 2-rank rehearsal (both ranks on GPU 0, gloo transport): the multi-rank control
flow the driver's `--gpus N` runs go through — process group, barriers, weak-scaled evaluation, the LM with its exchange statistics
and the strong-scaling LM section — at sizes that take seconds."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "3", "--warmup", "1", "--views", "40", "--grid", "20", "--c3-views", "16"]
FIELDS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
          "data", "config", "roofline", "cpu_baseline")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out  # exactly ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_on_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *COMMON], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert all(k in d for k in FIELDS), sorted(d)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["metric"] == "residual+Jacobian evals/sec" and d["dtype"] == "f64" and d["scaling"] == "weak" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(d["value"] - 40 * 400 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["lm"]["success"] and d["lm"]["allreduce"] == "none (1 rank)" and d["lm_strong"]["success"] and d["lm_strong"]["scaling"] == "strong"


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_on_one_gpu():
    env = dict(os.environ, CBA_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *COMMON, "--no-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)  # rank 0 alone prints
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    lm, ls = d["lm"], d["lm_strong"]
    assert lm["success"] and lm["views_total"] == 80 and "gloo" in lm["allreduce"]
    # one exchange point per LM step: the initial system, one per trial point, one more per miss / rejected step / plain-trial acceptance
    sp = lm["speculation"]
    assert lm["allreduce_calls"] >= 1 + lm["iterations"] and sp["speculative_steps"] >= 1
    assert ls["success"] and ls["scaling"] == "strong" and "split over 2 rank(s)" in ls["workload"] and ls["allreduce_calls"] >= 1 + ls["iterations"]
