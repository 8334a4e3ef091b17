#!/bin/bash
# Ablation of the MFMA filter (fp16 main-product form by default) (diagnostic builds -DGQHIP_ABL=<mask>: 1 no chunk barriers, 2 no chunk staging, 4 no
# candidate tracker, 8 no v_max3 fold): which part of the kernel separates it from its bare MFMA loop.  Results of the
# ablated builds are garbage by construction; only the filter kernel time is read.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
C=$REPO/vq-vae-from-gaussian-vae_amd/csrc
for round in 1 2; do
  for a in 0 1 2 3 4 8 12 15; do
    L=$C/libgqhip_abl$a.so; [ $a = 0 ] && L=$C/libgqhip.so
    [ -f $L ] || continue
    echo -n "abl=$a: "; GQHIP_LIB=$L timeout 120 python3 $REPO/tools/kbench.py --iters 20 2>&1 | tail -1 | cut -c1-170
  done
done
