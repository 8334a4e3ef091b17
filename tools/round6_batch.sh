#!/bin/bash
# One lease's evidence batch of round 6 (run through gpurun): the -m gpu suite with -x as the driver runs it, the per-round profile recipe
# (kernel stats + PMC passes), every BASELINE config's bench line (each with cpu_baseline + in-run parity), the module-level per-kernel
# breakdowns, the 400-seed stress sweep, and bench.py as ONE rank of an external launcher over RCCL (world size 1: the only size a one-GPU lease
# can run -- init, the packed all_gather_into_tensor on the device, barrier, teardown).  Everything lands in gpurun_out/r06_batch/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r06_batch
mkdir -p "$OUT"
cd "$R" || exit 1
export MIOPEN_USER_DB_PATH=$(mktemp -d)      # an empty MIOpen db, like the driver's fresh box
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > "$OUT/gpu_suite_empty_miopen_db.txt"
GQ_STRESS_SEEDS=400 python -m pytest tests/test_gpu_stress.py -x -q 2>&1 | tail -3 > "$OUT/stress400.txt"
tools/bench_all_configs.sh r06 > "$OUT/bench_all_configs.txt" 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 \
    --no-cpu-baseline --no-reference-gpu 2> "$OUT/rccl_world1_stderr.txt" | grep '^{' | tail -1 > "$OUT/line_rccl_world1.json"
tools/profile_round.sh r06 > "$OUT/profile_round_stdout.txt" 2>&1
tools/r6_modules_prof.sh > "$OUT/modules_prof_stdout.txt" 2>&1
tools/pmc_filter_dims.sh r06 > "$OUT/pmc_filter_dims_stdout.txt" 2>&1
ls "$OUT"
