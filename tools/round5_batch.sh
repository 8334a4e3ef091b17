mkdir -p gpurun_out/r05
(python -m pytest tests/test_gpu_grid.py -x -q 2>&1 | tail -3; for a in "--rows 65536 --dim 4" "--rows 16384 --dim 4"; do python tools/kbench.py $a; python tools/kbench.py $a; done) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/grid_concave.txt
cd /tmp && export TMPDIR=/tmp
for c in 1 0; do GQHIP_IMG_CACHE=$c rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05/prep_ab_$c -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --iters 40 > /dev/null 2>&1; done
cd $GRAFT_REPO_ROOT
for c in 1 0; do echo "GQHIP_IMG_CACHE=$c"; grep -h "gq_prep\|gq_filter_bf16\|gq_rerank" $(find gpurun_out/r05/prep_ab_$c -name '*kernel_stats.csv'); done > gpurun_out/r05/prep_image_cache_ab.txt
find gpurun_out/r05/prep_ab_0 gpurun_out/r05/prep_ab_1 -name "*.csv" -size +1M -delete
(GQ_STRESS_SEEDS=400 python -m pytest tests/test_gpu_stress.py -q -p no:cacheprovider 2>&1 | tail -3) > gpurun_out/r05/stress400.txt
(for ns in 8 16 64 256; do echo "--- GQHIP_SCORES_NSPLIT=$ns (row block 256 rows x $((65536/ns)) codes = $((65536/ns/1024)) page(s) of every row)"; GQHIP_SCORES_NSPLIT=$ns python tools/scores_bench.py --dims 16 --rows 16384 --iters 10 2>&1 | grep gq_scores | cut -c1-140; done) > gpurun_out/r05/scores_split_sweep.txt
bash tools/bench_all_configs.sh r05 > gpurun_out/r05/bench_all_configs.txt 2>&1
tail -12 gpurun_out/r05/bench_all_configs.txt
