"""Steady-state per-step kernel breakdown from a rocprofv3 --kernel-trace csv of bench.py:
steps are delimited by the launches of the step's filter kernel (gq_filter_bf16_kernel when the split-bf16 filter
runs, else gq_filter_kernel); the last SKIP of them are ignored (bench.py's untimed per-stage passes after the timed
region) and the K steps before those are summed.   usage: steady_profile.py TRACE_DIR [K=4] [SKIP=5]"""
import csv
import glob
import sys
from collections import defaultdict

d, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "gq_filter_bf16_kernel" in r["Kernel_Name"]]
if not marks:
    marks = [i for i, r in enumerate(rows) if "gq_filter_kernel" in r["Kernel_Name"]]
marks = marks[: len(marks) - skip] if skip else marks
lo, hi = marks[-last - 1], marks[-1]
acc, cnt = defaultdict(float), defaultdict(int)
for r in rows[lo:hi]:
    dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = r["Kernel_Name"][:100]
    acc[k] += dt
    cnt[k] += 1
span = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
tot = sum(acc.values())
print(f"{last} steady steps: wall span {span / last / 1e3:.2f} ms/step, kernel time {tot / last / 1e3:.2f} ms/step, "
      f"{sum(cnt.values()) / last:.0f} launches/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{v / last / 1e3:9.3f} ms/step {100 * v / tot:6.2f}%  x{cnt[k] / last:6.1f}  {k}")
