#!/bin/bash
# Round-2 measurement pass on the GPU box (via gpurun, from the repo root).  Outputs under gpurun_out/r02_*; the summaries that are
# cited in DESIGN.md are copied into profiles/ afterwards.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py > "$OUT/r02_bench.json" 2> "$OUT/r02_bench.err" || echo "bench failed"
python3 tools/exp_lm.py c1h c2 c3 c3e c5 > "$OUT/r02_lm.jsonl" 2> "$OUT/r02_lm.err" || echo "exp_lm failed"
python3 tools/exp_c5_fp32.py > "$OUT/r02_c5_fp32.json" 2> "$OUT/r02_c5_fp32.err" || echo "c5 fp32 failed"
python3 tools/fuzz_gpu.py 300 2027 > "$OUT/r02_fuzz_auto.json" 2> "$OUT/r02_fuzz_auto.err" || echo "fuzz failed"
CBA_LM_RESIDENT=0 python3 tools/fuzz_gpu.py 300 2028 > "$OUT/r02_fuzz_host.json" 2> "$OUT/r02_fuzz_host.err" || echo "fuzz host failed"
rm -f "$OUT/configs.jsonl"; python3 tools/bench_configs.py > "$OUT/r02_configs.log" 2>&1 || echo "configs failed"
python3 tools/bench_small.py > "$OUT/r02_small.log" 2>&1 || echo "small failed"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r02_prof_stats" -- python3 "$ROOT/bench.py" --steps 50 --warmup 5 --no-cpu > "$OUT/r02_bench_prof.json" 2> "$OUT/r02_bench_prof.err" || echo "stats run failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --no-lm > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.err" || echo "fetch run failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --no-lm > "$OUT/bench_write.json" 2> "$OUT/bench_write.err" || echo "write run failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/r02_prof_valu" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --no-lm > "$OUT/bench_valu.json" 2> "$OUT/bench_valu.err" || echo "valu run failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/r02_prof_valu2" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --no-lm > "$OUT/bench_valu2.json" 2> "$OUT/bench_valu2.err" || echo "valu2 run failed"
find "$OUT/r02_prof_stats" "$OUT/prof_fetch" "$OUT/prof_write" "$OUT/r02_prof_valu" -name "*.csv" | head -20
