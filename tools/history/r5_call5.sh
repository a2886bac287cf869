#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/stream_structure_probe.py 25.6 3 > gpurun_out/r5e_stream_structure.json 2>gpurun_out/r5e_stream_structure.err
cat gpurun_out/r5e_stream_structure.json; tail -3 gpurun_out/r5e_stream_structure.err
MV_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --pages 200000 --steps 30 --warmup 5 --cpu-sample-pages 1024 --no-aux > gpurun_out/r5e_bench_rccl1.txt 2>gpurun_out/r5e_bench_rccl1.err
tail -c 1900 gpurun_out/r5e_bench_rccl1.txt; tail -3 gpurun_out/r5e_bench_rccl1.err
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pinned_reference_fixture" > gpurun_out/r5e_fde_pin.log 2>&1; tail -3 gpurun_out/r5e_fde_pin.log
