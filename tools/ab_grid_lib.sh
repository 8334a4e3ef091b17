#!/bin/bash
# On the GPU box: A/B of two library builds on the dim-4 search -- rocprofv3 kernel-trace averages of tools/kbench.py at gq_1.00's shape,
# trained-like and flat rows, three interleaved rounds.  usage: tools/ab_grid_lib.sh libA.so libB.so
A=${1:-libgqhip.so}; B=${2:-libgqhip_noclamp.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do for L in $A $B; do for a in "--dim 4 --rows 65536" "--dim 4 --rows 65536 --flat"; do
  rm -rf /tmp/gr_prof; GQHIP_LIB=$R/vq-vae-from-gaussian-vae_amd/csrc/$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gr_prof -- python3 $R/tools/kbench.py --iters 40 $a > /tmp/gr_out.txt 2>&1
  python3 - "$L $a" $(find /tmp/gr_prof -name '*kernel_stats.csv') <<'PY'
import csv, sys
rows = {r["Name"].split("<")[0].split("::")[-1]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[2]))}
print(sys.argv[1], {k: round(v, 2) for k, v in rows.items() if k.startswith("gq_")})
PY
done; done; done
