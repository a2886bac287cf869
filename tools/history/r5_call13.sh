#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_store.py tests/test_gpu_sharded.py -x -q -m gpu -k "gpu_sharded_searcher or merge_of_gathered or comm or two_stage or bench_eight" > gpurun_out/r5r_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r5r_tests.log
MV_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --pages 125000 --steps 60 --warmup 5 --cpu-sample-pages 2048 --cpu-baseline-quick --no-aux > gpurun_out/r5r_bench_rccl1_125k.txt 2>gpurun_out/r5r_bench_rccl1.err; echo "rccl1 rc=$?"
tail -n 1 gpurun_out/r5r_bench_rccl1_125k.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'], 'step-kernel', round(d['ms_per_step']-d['roofline']['kernel_ms_avg'],4), d['config']['collective_and_merge_ms_per_step'], d['config']['step_minus_local_ms_per_step'], d['roofline']['launches_timed'])"
timeout 600 python bench.py --gpus 1 --pages 125000 --steps 60 --warmup 5 --cpu-sample-pages 2048 --cpu-baseline-quick --no-aux > gpurun_out/r5r_bench_plain_125k.txt 2>gpurun_out/r5r_bench_plain.err
tail -n 1 gpurun_out/r5r_bench_plain_125k.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('plain (no dist):', d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'], 'step-kernel', round(d['ms_per_step']-d['roofline']['kernel_ms_avg'],4))"
