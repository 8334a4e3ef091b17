#!/bin/bash
# Ablations of conv3x3_gn_f16x3_kernel (make -C vq-vae-from-gaussian-vae_amd/csrc ablu ABL=<mask>: 64 no staging of the next
# chunk, 128 weights loaded once, 256 one A operand per tap, 512 no epilogue) at the 256 x 256 level's shapes.
out=gpurun_out/abl_conv3.txt
: > $out
for a in 0 64 128 256 512 192 960; do
  lib=libgqhip_ablu$a.so; [ $a = 0 ] && lib=libgqhip.so
  echo "--- ABL=$a" >> $out
  GQHIP_LIB=$PWD/vq-vae-from-gaussian-vae_amd/csrc/$lib python tools/convstack/conv3_bench.py 2>&1 | grep "\^2:" | head -2 | cut -c1-160 >> $out
done
cat $out
