#!/bin/bash
# Power / clock of the device while one kernel loops (rocm-smi samples every 0.5 s): is a kernel spending the power budget?
# PROBE_DELAY=<s>: seconds before the first sample (default 3).
# usage: tools/convstack/power_probe.sh "<python command that runs ~8 s of the kernel>" ; writes gpurun_out/power_probe.txt
out=gpurun_out/power_probe.txt
echo "== $1" >> $out
bash -c "$1" > /dev/null 2>&1 &
pid=$!
sleep ${PROBE_DELAY:-3}
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|busy" | tr -s ' ' | tr '\n' ';' >> $out
  echo >> $out
  sleep 0.5
done
wait $pid
