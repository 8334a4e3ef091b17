#!/bin/bash
# Round 5's evidence batch on one lease (through gpurun): the -m gpu suite with an empty MIOpen db, the dim-4 search's kernel stats
# and PMC passes, the image-cache A/B of the first launch, the 400-seed stress, the compat op's split sweep, every BASELINE config's
# bench line.  Everything lands in gpurun_out/r05/ (copy what is judged into profiles/r05/).  usage: tools/round5_batch.sh <lease tag>
TAG=${1:-B}
export MIOPEN_USER_DB_PATH=$(mktemp -d) MIOPEN_CUSTOM_CACHE_DIR=$(mktemp -d)
mkdir -p gpurun_out/r05
(echo "lease $TAG: $(hostname) $(date -u +%F_%H:%M:%S) MIOPEN_USER_DB_PATH=$MIOPEN_USER_DB_PATH (empty)"; python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -12) > gpurun_out/r05/gpu_suite_lease_${TAG}_empty_miopen_db.txt 2>&1
tail -3 gpurun_out/r05/gpu_suite_lease_${TAG}_empty_miopen_db.txt
bash tools/pmc_grid.sh r05 dim4 > /dev/null 2>&1
bash tools/pmc_grid_insts.sh r05 > gpurun_out/r05/pmc_grid_insts.txt 2>&1
(export GQHIP_LIB=$PWD/vq-vae-from-gaussian-vae_amd/csrc/libgqhip_stamps.so; python tools/grid_phases.py; python tools/grid_phases.py --flat) 2>&1 | grep -v amdgpu > gpurun_out/r05/grid_phases.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in 1 0; do GQHIP_IMG_CACHE=$c rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/prep_ab_$c -- python3 $R/tools/kbench.py --iters 40 > /dev/null 2>&1; done
cd $R
for c in 1 0; do echo "GQHIP_IMG_CACHE=$c"; grep -h "gq_prep\|gq_filter_bf16\|gq_rerank" $(find gpurun_out/r05/prep_ab_$c -name '*kernel_stats.csv'); done > gpurun_out/r05/prep_image_cache_ab.txt
find gpurun_out/r05/prep_ab_0 gpurun_out/r05/prep_ab_1 -name "*.csv" -size +1M -delete
(GQ_STRESS_SEEDS=400 python -m pytest tests/test_gpu_stress.py -q -p no:cacheprovider 2>&1 | tail -3) > gpurun_out/r05/stress400.txt
(for ns in 8 16 64 256; do echo "--- GQHIP_SCORES_NSPLIT=$ns (row block: 256 rows x $((65536/ns)) codes)"; GQHIP_SCORES_NSPLIT=$ns python tools/scores_bench.py --dims 16 --rows 16384 --iters 10 2>&1 | grep gq_scores | cut -c1-140; done) > gpurun_out/r05/scores_split_sweep.txt
(for a in "--rows 65536 --dim 4" "--rows 65536 --dim 4 --flat" "--rows 16384 --dim 4" "--rows 16384 --dim 16" "--rows 32768 --dim 8" "--rows 65536 --dim 16 --vq"; do python tools/kbench.py $a 2>&1 | grep -v amdgpu; done) > gpurun_out/r05/kbench_lines.txt
bash tools/bench_all_configs.sh r05 > gpurun_out/r05/bench_all_configs.txt 2>&1
tail -9 gpurun_out/r05/bench_all_configs.txt
