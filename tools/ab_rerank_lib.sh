#!/bin/bash
# On the GPU box (through gpurun): A/B of two library builds on the re-rank kernel -- rocprofv3 kernel-trace average of gq_rerank_kernel
# and of the filter, whole-call wall of tools/kbench.py, two interleaved rounds.  usage: tools/ab_rerank_lib.sh libA.so libB.so
A=${1:-libgqhip.so}; B=${2:-libgqhip_old.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do for L in $A $B; do for a in "--dim 16 --rows 16384" "--dim 8 --rows 32768" "--dim 16 --rows 65536"; do
  rm -rf /tmp/rr_prof; GQHIP_LIB=$R/vq-vae-from-gaussian-vae_amd/csrc/$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rr_prof -- python3 $R/tools/kbench.py --iters 40 $a > /tmp/rr_out.txt 2>&1
  python3 - "$L $a" $(find /tmp/rr_prof -name '*kernel_stats.csv') <<'PY'
import csv, sys
rows = {r["Name"].split("<")[0].split("::")[-1]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[2]))}
print(sys.argv[1], {k: round(v, 2) for k, v in rows.items() if k.startswith("gq_")})
PY
done; done; done
