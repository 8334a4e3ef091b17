#!/bin/bash
# Run on the GPU box: kernel trace of bench.py and the steady-state per-kernel table only (the quick half of
# tools/profile_round.sh).  usage: tools/convstack/profile_step.sh <tag>
set -u
R=${1:-step}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 "$REPO/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OUT/bench_stdout.txt" 2>&1
python3 "$REPO/tools/steady_profile.py" "$OUT/bench_trace" 4 28 > "$OUT/STEADY_STATE.txt" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
cut -c1-170 "$OUT/STEADY_STATE.txt" | head -45
