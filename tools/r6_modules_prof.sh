#!/bin/bash
# On the GPU box: per-kernel breakdown of the module-level eval forwards (rocprofv3 --kernel-trace --stats) -> gpurun_out/r6_mod/
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r6_mod
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -- python3 "$R/tools/mbench.py" "$@" > "$OUT/$tag.txt" 2>&1
  f=$(find "$OUT/$tag" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${tag}_kernel_stats.csv"
  find "$OUT/$tag" -name "*.csv" -size +1M -delete
}
prof gq2_256 --module gq2 --dim 16 --bs 16 --size 256
prof vq_512 --module vq --dim 16 --bs 16 --size 512
prof vq_256 --module vq --dim 16 --bs 16 --size 256
prof gq_256 --module gq --dim 16 --bs 16 --size 256
cd "$R"
for t in gq2_256 vq_512 vq_256 gq_256; do echo "== $t"; tail -1 "$OUT/$t.txt"; head -12 "$OUT/${t}_kernel_stats.csv" | cut -c1-200; done
