#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + PMC passes of the quantiser microbench at the OTHER BASELINE shapes --
# gq_1.00 (dim 4, 65 536 rows: packed split-bf16 filter), gq_0.50 (dim 8, 32 768 rows: fp16 main product) and vq_16 at 512 x 512
# (dim 16, 65 536 rows, A = -1).  One counter set per pass (MI355X_MICROARCH.md: separate --pmc passes, no trace domains besides
# --kernel-trace).  Summary -> gpurun_out/pmc_dims_<tag>/SUMMARY.txt; copy it to profiles/.
set -u
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_dims_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_shape() {   # name, kbench args...
  local name=$1; shift
  local D=$OUT/$name
  mkdir -p "$D"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$D/kbench_trace" -- python3 "$REPO/tools/kbench.py" --iters 20 "$@" > "$D/kbench_stdout.txt" 2>&1
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo "$C" | tr ' ' '_' | cut -c1-48)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D/pmc_$tag" -- python3 "$REPO/tools/kbench.py" --iters 5 "$@" > "$D/pmc_${tag}_stdout.txt" 2>&1
  done
  python3 "$REPO/tools/summarize_prof.py" "$D" > "$D/SUMMARY.txt" 2>&1
  find "$D" -name "*.csv" -size +1M -delete
}
run_shape gq_1.00_dim4  --dim 4 --rows 65536
run_shape gq_0.50_dim8  --dim 8 --rows 32768
run_shape vq_16_512     --dim 16 --rows 65536 --vq
run_shape gq_0.25_dim16 --dim 16 --rows 16384
cat "$OUT"/*/SUMMARY.txt > "$OUT/SUMMARY_all.txt"
grep -h "filter kernel" "$OUT"/*/kbench_stdout.txt
