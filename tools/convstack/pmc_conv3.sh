#!/bin/bash
# PMC passes on the direct convolution kernels (tools/convstack/conv3_bench.py), one rocprofv3 --pmc run per counter group (never
# combined with trace domains other than --kernel-trace).  Run on the GPU box; writes gpurun_out/pmc_conv3_summary.txt.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_c3
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo "$C" | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c3/$tag -- python3 $R/tools/convstack/conv3_bench.py > $R/gpurun_out/pmc_c3_stdout.txt 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/pmc_c3/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3" in k:
            acc[k[:70] + " | grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = ["PMC passes on tools/convstack/conv3_bench.py (tools/convstack/pmc_conv3.sh: separate rocprofv3 --pmc runs; averages per launch).",
       "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES = 32 x MFMA count, GRBM_GUI_ACTIVE sums the 8 XCDs;",
       "FETCH_SIZE / WRITE_SIZE in KiB (FETCH to be doubled on gfx950).  Launches of equal grid size (different Cin) are averaged together.", ""]
for k, v in sorted(acc.items()):
    out.append(k)
    out.append("    " + ", ".join(f"{c}={sum(x) / len(x):.4g}" for c, x in sorted(v.items())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        busy = (sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"])) / (1024 * (sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"])) / 8)
        out.append(f"    -> matrix pipes busy {100 * busy:.1f} % of the kernel's SIMD-cycles")
open(R + "/gpurun_out/pmc_conv3_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
