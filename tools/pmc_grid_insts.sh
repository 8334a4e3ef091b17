#!/bin/bash
# Run on the GPU box (through gpurun): instruction-mix PMC pass of the dim-4 search (csrc/gq_grid.h) at the trained operating point
# and on sigma ~ 1 rows (kbench --flat).  Prints per-launch SQ counters of gq_grid_kernel.  usage: tools/pmc_grid_insts.sh [tag]
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_grid_insts_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for regime in trained flat; do
  extra=""; [ $regime = flat ] && extra="--flat"
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo "$C" | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$regime/$tag" -- python3 "$REPO/tools/kbench.py" --iters 3 --dim 4 --rows 65536 $extra > /dev/null 2>&1
  done
  python3 - "$OUT/$regime" $regime <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gq_grid_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
done
find "$OUT" -name "*.csv" -size +1M -delete
