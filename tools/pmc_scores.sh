#!/bin/bash
# PMC passes on the compat op's score kernel (tools/scores_bench.py at 16 384 x 65 536 x dim 16): one rocprofv3 --pmc run per
# counter group (never combined with trace domains other than --kernel-trace), for the shipped kernel and for the two diagnostic
# builds (no matrix work = libgqhip_abl16.so, no stores = libgqhip_abl32.so: `make abl ABL=16`, `make abl ABL=32`).
# Writes gpurun_out/pmc_scores_summary.txt.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_sc
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_WR[A-Z_0-9]*\|TCC_[A-Z_0-9]*STALL[A-Z_0-9]*\|TCP_[A-Z_0-9]*STALL[A-Z_0-9]*\|TCP_TCC_WRITE[A-Z_0-9]*\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_WAIT_INST_[A-Z]*\|TA_[A-Z_0-9]*STALL[A-Z_0-9]*" | sort -u > $R/gpurun_out/pmc_scores_names.txt
for lib in libgqhip.so libgqhip_abl16.so libgqhip_abl32.so; do
  export GQHIP_LIB=$R/vq-vae-from-gaussian-vae_amd/csrc/$lib
  gi=0
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "WRITE_SIZE"; do
    gi=$((gi+1))
    tag=${lib}_g$gi
    timeout -s KILL 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sc/$tag -- python3 $R/tools/scores_bench.py --dims 16 --rows 16384 --iters 4 > $R/gpurun_out/pmc_sc_stdout_$lib.txt 2>&1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(R + "/gpurun_out/pmc_sc/**/*counter_collection.csv", recursive=True):
    lib = f.split("pmc_sc/")[1].split(".so")[0]
    for r in csv.DictReader(open(f)):
        if "gq_scores" in r["Kernel_Name"]:
            acc[lib][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(R + "/gpurun_out/pmc_sc/**/*kernel_trace.csv", recursive=True):
    lib = f.split("pmc_sc/")[1].split(".so")[0]
    if "_g1/" not in f:      # the pass that also counted GRBM_GUI_ACTIVE
        continue
    for r in csv.DictReader(open(f)):
        if "gq_scores" in r["Kernel_Name"]:
            dur[lib].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = ["PMC passes on the compat op's kernel (dim 16: gq_scores_f16x3_kernel<16, 2, 8>) at 16 384 x 65 536 (tools/pmc_scores.sh; separate rocprofv3 --pmc runs; averages per launch).",
       "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave, SQ_VALU_MFMA_BUSY_CYCLES sums SIMD-cycles, GRBM_GUI_ACTIVE sums the 8 XCDs;",
       "WRITE_SIZE in KiB.  Durations (us) are those of the pass that also counted GRBM_GUI_ACTIVE.", ""]
for lib, v in sorted(acc.items()):
    d = sum(dur[lib]) / len(dur[lib]) if dur[lib] else None
    out.append(lib + ".so" + (f"   kernel duration under PMC: {d:.0f} us" if d else ""))
    out.append("    " + ", ".join(f"{c}={sum(x) / len(x):.4g}" for c, x in sorted(v.items())))
    g = lambda c: sum(v[c]) / len(v[c]) if c in v else None
    if g("GRBM_GUI_ACTIVE") and d:
        out.append(f"    -> clock {g('GRBM_GUI_ACTIVE') / 8 / d / 1e3:.2f} GHz averaged over the kernel")
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
        out.append(f"    -> matrix pipes busy {100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * g('GRBM_GUI_ACTIVE') / 8):.1f} % of the kernel's SIMD-cycles")
    if g("SQ_WAIT_INST_ANY") and g("SQ_WAVE_CYCLES"):
        out.append(f"    -> SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {100 * g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.1f} %")
open(R + "/gpurun_out/pmc_scores_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
print(open(R + "/gpurun_out/pmc_scores_names.txt").read())
PY
