#!/bin/bash
# A/B of the filter's fold pipeline depth (GQHIP_FILTER_PD=1: round 3's, 2: round 4's) on the quantiser microbench, interleaved.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for args in "--dim 16 --rows 16384" "--dim 16 --rows 65536" "--dim 8 --rows 32768" "--dim 32 --rows 16384" "--dim 16 --rows 65536 --vq"; do
    for pd in 1 2; do
      echo -n "PD=$pd  "
      GQHIP_FILTER_PD=$pd python3 "$REPO/tools/kbench.py" --iters 30 $args 2>&1 | tail -1
    done
  done
done
