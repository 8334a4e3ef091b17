#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of bench.py, then PMC passes on the
# quantiser-only microbench (same filter kernel, same shape).  Summaries land in gpurun_out/.
set -u
R=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 "$REPO/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-reference-gpu > "$OUT/bench_stdout.txt" 2>&1
# after the timed region bench.py runs 5 (stage split) + 23 (back-to-back quantiser calls) more split-bf16 filter launches
python3 "$REPO/tools/steady_profile.py" "$OUT/bench_trace" 4 28 > "$OUT/STEADY_STATE.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kbench_trace" -- python3 "$REPO/tools/kbench.py" --iters 20 > "$OUT/kbench_stdout.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kbench_fp32_trace" -- python3 "$REPO/tools/kbench.py" --iters 20 --filter fp32 > "$OUT/kbench_fp32_stdout.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/variants_trace" -- python3 "$REPO/tools/kbench_variants.py" > "$OUT/variants_stdout.txt" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo "$C" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$tag" -- python3 "$REPO/tools/kbench.py" --iters 5 > "$OUT/pmc_${tag}_stdout.txt" 2>&1
done
# compact summaries
python3 "$REPO/tools/summarize_prof.py" "$OUT" > "$OUT/SUMMARY.txt" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
ls -R "$OUT" | head -50
