#!/bin/bash
# Copy the judged summaries of a tools/profile_round.sh run from gpurun_out/prof_<tag>/ into profiles/<round>/.
# usage: tools/collect_profiles.sh r01g r01
set -eu
SRC=gpurun_out/prof_$1
DST=profiles/${2:-r01}
mkdir -p "$DST"
cp "$SRC/SUMMARY.txt" "$SRC/STEADY_STATE.txt" "$DST/"
cp "$(find "$SRC/bench_trace" -name '*kernel_stats.csv' | head -1)" "$DST/bench_kernel_stats.csv"
cp "$(find "$SRC/kbench_trace" -name '*kernel_stats.csv' | head -1)" "$DST/kbench_kernel_stats.csv"
cp "$(find "$SRC/kbench_fp32_trace" -name '*kernel_stats.csv' | head -1)" "$DST/kbench_fp32_kernel_stats.csv"
if [ -d "$SRC/variants_trace" ]; then   # compat score op, VQ / LFQ at 512x512, module-level call (tools/kbench_variants.py)
  cp "$(find "$SRC/variants_trace" -name '*kernel_stats.csv' | head -1)" "$DST/variants_kernel_stats.csv"
  grep -h -E "^(compat|vq_|lfq_|gq_)" "$SRC/variants_stdout.txt" > "$DST/variants_lines.txt" || true
fi
for C in FETCH_SIZE WRITE_SIZE; do
  # keep only the gqhip kernels' rows (small, what bench.py reads)
  f=$(find "$SRC/pmc_$C" -name '*counter_collection.csv' | head -1)
  (head -1 "$f"; grep "gqhip::" "$f") > "$DST/pmc_$C.csv"
done
grep -h '^{' "$SRC/bench_stdout.txt" | tail -1 > "$DST/bench_line_under_rocprof.json"
grep -h "filter kernel" "$SRC/kbench_stdout.txt" | tail -1 > "$DST/kbench_line.txt"
grep -h "filter kernel" "$SRC/kbench_fp32_stdout.txt" | tail -1 > "$DST/kbench_fp32_line.txt"
ls -la "$DST"
