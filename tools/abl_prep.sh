#!/bin/bash
# Ablation of gq_prep_kernel (diagnostic builds `make -C vq-vae-from-gaussian-vae_amd/csrc abl ABL=<mask>`: 1024 code blocks do nothing,
# 2048 row blocks do nothing, 4096 fp32 exp / log instead of fp64): what its ~10 us are made of.  Results of the ablated builds are
# garbage by construction; only the kernel's average duration (rocprofv3 --kernel-trace --stats) is read.  Run on the GPU box.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
C=$REPO/vq-vae-from-gaussian-vae_amd/csrc
cd /tmp && export TMPDIR=/tmp
for a in 0 1024 2048 4096 3072; do
  L=$C/libgqhip_abl$a.so; [ $a = 0 ] && L=$C/libgqhip.so
  [ -f $L ] || continue
  export GQHIP_LIB=$L
  for mode in rows z; do
    rm -rf /tmp/ablp
    if [ $mode = rows ]; then
      rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ablp -- python3 $REPO/tools/kbench.py --iters 20 > /dev/null 2>&1
    else
      rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ablp -- python3 $REPO/tools/mbench.py --module gq --iters 20 > /dev/null 2>&1
    fi
    f=$(find /tmp/ablp -name '*kernel_stats.csv' | head -1)
    echo "abl=$a ($mode): $(grep gq_prep_kernel $f | awk -F, '{printf "%s calls, avg %.2f us, min %.2f us; ", $(NF-6), $(NF-4)/1000, $(NF-2)/1000}')"
  done
done
