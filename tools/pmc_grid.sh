#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate PMC passes (MI355X_MICROARCH.md: one counter set per pass, no
# trace domain besides --kernel-trace) of the quantiser microbench at the grid-search shapes (csrc/gq_grid.h): gq_1.00 (dim 4,
# 65 536 rows) and gq_0.50 (dim 8, 32 768 rows).  Summary -> gpurun_out/pmc_grid_<tag>/SUMMARY.txt; copy it to profiles/.
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_grid_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_shape() {
  local name=$1; shift
  local D=$OUT/$name
  mkdir -p "$D"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 "$REPO/tools/kbench.py" --iters 20 "$@" > "$D/kbench_stdout.txt" 2>&1
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo "$C" | tr ' ' '_' | cut -c1-48)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D/pmc_$tag" -- python3 "$REPO/tools/kbench.py" --iters 5 "$@" > "$D/pmc_${tag}_stdout.txt" 2>&1
  done
  python3 "$REPO/tools/summarize_prof.py" "$D" > "$D/SUMMARY.txt" 2>&1
  find "$D" -name "*.csv" -size +1M -delete
}
run_shape gq_1.00_dim4 --dim 4 --rows 65536
[ "${2:-}" = "dim4" ] || run_shape gq_0.50_dim8 --dim 8 --rows 32768
cat "$OUT"/*/SUMMARY.txt > "$OUT/SUMMARY_all.txt"
grep -h "kernel" "$OUT"/*/kbench_stdout.txt
