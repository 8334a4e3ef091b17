#!/bin/bash
# A/B of two builds of libgqhip.so on the same box, interleaved rounds of tools/kbench.py.
# usage: tools/calibration/ab_lib.sh <libA.so> <libB.so> [rounds] [kbench args...]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; R=${3:-3}; shift 3
for round in $(seq 1 $R); do
  for L in $A $B; do
    echo -n "$(basename $L): "; GQHIP_LIB=$REPO/$L timeout 120 python3 $REPO/tools/kbench.py --iters 30 "$@" 2>&1 | tail -1 | cut -c1-210
  done
done
