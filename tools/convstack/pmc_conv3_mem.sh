#!/bin/bash
# Memory-pipeline counters of conv3x3_gn_f16x3_kernel<1,128> (16 x 256 x 256 x 128 -> 128): texture-addresser / L1 / L2 stall and
# hit counters, one rocprofv3 --pmc pass per group (at most two
# counters of a block per pass: more "exceeds the capabilities of the hardware" and rocprofv3 then hangs; each pass under timeout), for the shipped kernel and the "weights loaded once" / "no staging"
# ablation builds.  Writes gpurun_out/pmc_conv3_mem.txt.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_c3m
for a in 0 128 64; do
  lib=libgqhip_ablu$a.so; [ $a = 0 ] && lib=libgqhip.so
  export GQHIP_LIB=$R/vq-vae-from-gaussian-vae_amd/csrc/$lib
  i=0
  for C in "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" ; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c3m/abl${a}_g$i -- python3 $R/tools/convstack/c3_timeline.py 128 nostamps > $R/gpurun_out/pmc_c3m_stdout_${a}_$i.txt 2>&1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
out = ["conv3x3_gn_f16x3_kernel<1,128>, 16 x 256 x 256 x 128 -> 128: memory-pipeline counters per launch (tools/convstack/pmc_conv3_mem.sh)", ""]
for a in (0, 128, 64):
    acc = collections.defaultdict(list)
    for f in glob.glob(R + f"/gpurun_out/pmc_c3m/abl{a}_g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3x3_gn_f16x3" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.append(f"ABL={a}: " + ", ".join(f"{k}={sum(v) / len(v):.4g}" for k, v in sorted(acc.items())))
open(R + "/gpurun_out/pmc_conv3_mem.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
grep -l -i "error\|invalid\|not found" $R/gpurun_out/pmc_c3m_stdout_*.txt | head
