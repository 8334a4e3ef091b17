#!/bin/bash
# On the GPU box (through gpurun): bench.py lines of every BASELINE config -> gpurun_out/lines_<tag>/line_*.json
# (copy the ones to keep into profiles/rNN/).  usage: tools/bench_all_configs.sh r04
set -u
R=${1:-r01}
OUT=gpurun_out/lines_$R
mkdir -p "$OUT"
run() { name=$1; shift; python bench.py "$@" 2>/dev/null | grep '^{' | tail -1 > "$OUT/line_$name.json"; python - "$OUT/line_$name.json" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read()); r = l["roofline"]
print(sys.argv[1].split("/")[-1], l["value"], l["unit"], "vs_baseline", l.get("vs_baseline"), "frac", r.get("frac"), "filter us", r.get("avg_launch_us"),
      "cpu_baseline", (l.get("cpu_baseline") or {}).get("value"), "within_gates", l.get("parity", {}).get("within_gates"))
PY
}
run gq_0.25
run gq_0.50 --config gq_0.50
run gq_1.00 --config gq_1.00
run gq2_0.25 --config gq2_0.25
run gq_0.25_512 --size 512
run vq_16_512 --config vq_16 --size 512
run lfq_16_512 --config lfq_16 --size 512
run gq_0.25_50steps --steps 50 --warmup 5
