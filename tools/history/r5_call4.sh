#!/bin/bash
# round 5, call 4: where the FDE scan's transport stands (stream-only forms), the one-collective N>1 step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 300 python tools/scan_ceiling_probe.py 25.6 5 > gpurun_out/r5d_scan_ceiling.jsonl 2>gpurun_out/r5d_ceiling.err
MV_FDE_SCAN_PPW=8 timeout 300 python tools/scan_ceiling_probe.py 25.6 5 >> gpurun_out/r5d_scan_ceiling.jsonl 2>>gpurun_out/r5d_ceiling.err
MV_FDE_SCAN_BLOCKS_PER_CU=1 timeout 300 python tools/scan_ceiling_probe.py 25.6 5 >> gpurun_out/r5d_scan_ceiling.jsonl 2>>gpurun_out/r5d_ceiling.err
cat gpurun_out/r5d_scan_ceiling.jsonl
timeout 900 python -m pytest tests/test_gpu_store.py tests/test_gpu_sharded.py -x -q -m gpu -k "merge_of_gathered or gpu_sharded_searcher or bench_two_ranks" > gpurun_out/r5d_sharded_tests.log 2>&1
echo "sharded tests rc=$?"; tail -5 gpurun_out/r5d_sharded_tests.log
MV_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --backend gloo --pages 400000 --steps 20 --warmup 5 --cpu-sample-pages 1024 --no-aux > gpurun_out/r5d_bench_2rank_gloo.txt 2>gpurun_out/r5d_bench_2rank_gloo.err
tail -c 2500 gpurun_out/r5d_bench_2rank_gloo.txt
MV_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --pages 200000 --steps 30 --warmup 5 --cpu-sample-pages 1024 --no-aux > gpurun_out/r5d_bench_rccl1.txt 2>gpurun_out/r5d_bench_rccl1.err
tail -c 2500 gpurun_out/r5d_bench_rccl1.txt; tail -3 gpurun_out/r5d_bench_rccl1.err
