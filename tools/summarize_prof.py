"""Condense rocprofv3 csv output (kernel stats + PMC rows) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name, n=90):
    return name if len(name) <= n else name[:n] + "..."


for tag in ("bench_trace", "kbench_trace", "kbench_fp32_trace", "variants_trace", "trace"):
    files = glob.glob(os.path.join(out, tag, "**", "*kernel_stats.csv"), recursive=True)
    print(f"== {tag}: kernel stats (top 25 by total time) ==")
    for f in files:
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("TotalDuration", 0)) or 0))
        tot = sum(float(r.get("TotalDurationNs", 0) or 0) for r in rows)
        print(f"total kernel time {tot / 1e6:.2f} ms over {len(rows)} distinct kernels")
        for r in rows[:25]:
            print(f"{float(r['TotalDurationNs']) / 1e6:10.3f} ms  {float(r['Percentage']):6.2f}%  calls {r['Calls']:>6}  "
                  f"avg {float(r['AverageNs']) / 1e3:10.2f} us  {short(r['Name'])}")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "gq_" in k or "rerank" in k or "exhaustive" in k:
                acc[short(k, 60)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {os.path.basename(d)} ==")
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"  {k}: {c} avg/launch {sum(v) / len(v):.4g} over {len(v)} launches")
