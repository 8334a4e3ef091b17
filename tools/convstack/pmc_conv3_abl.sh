#!/bin/bash
# Clock and matrix-pipe share of conv3x3_gn_f16x3_kernel<1,128> (16 x 256 x 256 x 128 -> 128) for the shipped kernel and its
# ablation builds (make -C vq-vae-from-gaussian-vae_amd/csrc ablu ABL=<mask>): one rocprofv3 --pmc pass each (GRBM_GUI_ACTIVE,
# SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY) with --kernel-trace for the durations.
# Writes gpurun_out/pmc_conv3_abl.txt.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_c3a
for a in 0 64 128 256 512 192 960; do
  lib=libgqhip_ablu$a.so; [ $a = 0 ] && lib=libgqhip.so
  export GQHIP_LIB=$R/vq-vae-from-gaussian-vae_amd/csrc/$lib
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv \
    -d $R/gpurun_out/pmc_c3a/abl$a -- python3 $R/tools/convstack/c3_timeline.py 128 nostamps > $R/gpurun_out/pmc_c3a_stdout.txt 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
out = ["conv3x3_gn_f16x3_kernel<1,128>, 16 x 256 x 256 x 128 -> 128 with residual + statistics: clock (GRBM_GUI_ACTIVE / 8 / duration) and",
       "matrix-pipe share (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)) of the shipped kernel and its ablation builds", ""]
for d in sorted(glob.glob(R + "/gpurun_out/pmc_c3a/abl*"), key=lambda s: int(s.split("abl")[-1])):
    acc = collections.defaultdict(list)
    dur = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3x3_gn_f16x3" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3x3_gn_f16x3" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    g = {k: sum(v) / len(v) for k, v in acc.items()}
    t = sum(dur) / len(dur)
    cyc = g["GRBM_GUI_ACTIVE"] / 8
    out.append(f"ABL={d.split('abl')[-1]:>4s}: {t:7.1f} us  {cyc / 1e6:.3f} M cycles  clock {cyc / t / 1e3:.2f} GHz  matrix pipes busy {100 * g['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.1f} %"
               f"  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES {100 * g['SQ_WAIT_INST_ANY'] / g['SQ_WAVE_CYCLES']:.1f} %")
open(R + "/gpurun_out/pmc_conv3_abl.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
