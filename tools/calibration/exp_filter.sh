#!/bin/bash
# Filter-kernel experiments, interleaved rounds (cdna_hip_programming.md rule 24; separate processes: the knobs are read once).
OUT=${1:-gpurun_out/exp}
mkdir -p $OUT
run() { echo "== $*" >> $OUT/kbench_variants.txt; env "$@" python tools/kbench.py --iters 100 2>/dev/null | grep rows= >> $OUT/kbench_variants.txt; }
for rep in 1 2 3; do
  run A=default
  run GQHIP_BF16_GT=1
  run GQHIP_BF16_GT=2
  run GQHIP_BF16_GT=8
done
