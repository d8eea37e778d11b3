#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats + separate PMC passes for
# the bench workload.  Outputs under gpurun_out/prof_*; copy summaries into profiles/ afterwards.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
STEPS="${STEPS:-50}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- python3 "$ROOT/bench.py" --steps "$STEPS" --warmup 5 --no-cpu > "$OUT/bench_prof.json" 2> "$OUT/bench_prof.err" || echo "stats run failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --no-lm > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.err" || echo "fetch run failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --no-lm > "$OUT/bench_write.json" 2> "$OUT/bench_write.err" || echo "write run failed"
find "$OUT" -name "*.csv" | head -30
