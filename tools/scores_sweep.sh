#!/bin/bash
# A/B of the compat op's write-pattern switches: chunk / pair rotation (GQHIP_SCORES_ROT bits 0 / 1) and code splits
# (GQHIP_SCORES_NSPLIT), with and without the matrix work (libgqhip_abl16.so = `make abl ABL=16`).
# Output: gpurun_out/scores_sweep.txt
out=gpurun_out/scores_sweep.txt
: > $out
for lib in libgqhip.so libgqhip_abl16.so; do
  for rot in 0 1 2 3; do
    for ns in 8 16 64 256; do
      echo "--- $lib rot=$rot nsplit=$ns" >> $out
      GQHIP_LIB=$PWD/vq-vae-from-gaussian-vae_amd/csrc/$lib GQHIP_SCORES_NSPLIT=$ns GQHIP_SCORES_ROT=$rot \
        python tools/scores_bench.py --dims 16 --rows 16384 --iters 10 2>&1 | grep gq_scores | cut -c1-120 >> $out
    done
  done
done
cat $out
