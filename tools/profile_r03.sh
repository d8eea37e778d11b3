#!/bin/bash
# Round-3 measurement pass on the GPU box (via gpurun, from the repo root).  Outputs under gpurun_out/r03_*; the summaries cited in
# DESIGN.md are copied into profiles/ afterwards.  usage: bash tools/profile_r03.sh [part ...]   parts: evalpmc bench lm trace modebpmc
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
PARTS="${*:-evalpmc bench lm trace}"
cd "$ROOT"
for part in $PARTS; do
case $part in
evalpmc)
  for shape in c3 c4; do
    for seg in 1 0; do
      CBA_EVAL_SEGMENTS=$seg python3 tools/eval_only.py $shape 5 >> "$OUT/r03_eval_shapes.jsonl" 2>> "$OUT/r03_eval_shapes.err" || echo "eval $shape failed"
    done
    (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/r03_pmc_fetch_$shape" -- python3 "$ROOT/tools/eval_only.py" $shape 3 > "$OUT/r03_pmc_fetch_$shape.log" 2>&1) || echo "fetch $shape failed"
    (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/r03_pmc_write_$shape" -- python3 "$ROOT/tools/eval_only.py" $shape 3 > "$OUT/r03_pmc_write_$shape.log" 2>&1) || echo "write $shape failed"
  done
  python3 tools/eval_only.py c2 20 >> "$OUT/r03_eval_shapes.jsonl" 2>> "$OUT/r03_eval_shapes.err"
  tail -6 "$OUT/r03_eval_shapes.jsonl"
  ;;
bench)
  python3 bench.py > "$OUT/r03_bench.json" 2> "$OUT/r03_bench.err" || echo "bench failed rc=$?"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r03_prof_stats" -- python3 "$ROOT/bench.py" --steps 50 --warmup 5 --no-cpu > "$OUT/r03_bench_prof.json" 2> "$OUT/r03_bench_prof.err") || echo "stats run failed"
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --no-lm > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.err") || echo "fetch run failed"
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --no-lm > "$OUT/bench_write.json" 2> "$OUT/bench_write.err") || echo "write run failed"
  f=$(find "$OUT/r03_prof_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/r03_bench_kernel_stats.csv"
  head -c 600 "$OUT/r03_bench.json"; echo
  ;;
lm)
  for ctl in 1 0; do
    CBA_LM_CTL=$ctl CBA_LM_TIMING=1 python3 tools/exp.py lm c1h c2 c3 c3e c5 >> "$OUT/r03_lm_ctl$ctl.jsonl" 2>> "$OUT/r03_lm_ctl$ctl.err" || echo "exp lm failed"
  done
  grep -h shape "$OUT/r03_lm_ctl1.jsonl" | cut -c1-140
  ;;
modebpmc)
  # issue / stall split of the Mode B kernels (C2 direct form in bench.py's mode_b section, C3 moment form in lm_strong): two counter
  # passes, nothing but --pmc (tools/pmc_valu_summary.py r03 -> profiles/r03_pmc_valu_mode_a_b.json)
  rm -rf "$OUT/r03_prof_valu" "$OUT/r03_prof_valu2"
  (cd /tmp && rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/r03_prof_valu" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu > "$OUT/r03_bench_valu.json" 2> "$OUT/r03_bench_valu.err") || echo "valu run failed"
  (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/r03_prof_valu2" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu > "$OUT/r03_bench_valu2.json" 2> "$OUT/r03_bench_valu2.err") || echo "valu2 run failed"
  python3 tools/pmc_valu_summary.py r03 && rm -rf "$OUT/r03_prof_valu" "$OUT/r03_prof_valu2" || echo "summary failed (raw counter files kept)"
  ;;
trace)
  rm -rf "$OUT/r03_prof_c3e"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r03_prof_c3e" -- python3 "$ROOT/tools/exp.py" lm c3e > "$OUT/r03_prof_c3e.log" 2>&1) || echo "trace failed"
  f=$(find "$OUT/r03_prof_c3e" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/r03_c3e_lm_kernel_stats.csv"
  t=$(find "$OUT/r03_prof_c3e" -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/timeline.py "$t" 3 > "$OUT/r03_c3e_step_timeline.txt" && cat "$OUT/r03_c3e_step_timeline.txt"
  ;;
esac
done
