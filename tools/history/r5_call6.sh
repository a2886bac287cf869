#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_store.py -x -q -m gpu -k "fde_scan_ldsdma or fde_coarse_scan_and_pipeline or gpu_sharded_searcher or fde_batched_coarse_scan_matches" > gpurun_out/r5f_tests.log 2>&1
echo "tests rc=$?" ; tail -4 gpurun_out/r5f_tests.log
timeout 300 python tools/scan_ceiling_probe.py 25.6 5 > gpurun_out/r5f_scan_ceiling.jsonl 2>gpurun_out/r5f_ceiling.err; cat gpurun_out/r5f_scan_ceiling.jsonl
MV_PROBE_SWEEP=2 timeout 600 python tools/stream_structure_probe.py 25.6 3 > gpurun_out/r5f_stream_structure2.json 2>gpurun_out/r5f_stream_structure2.err
python -c "
import json; d=json.load(open('gpurun_out/r5f_stream_structure2.json'))
for k,v in d.items(): print(k, v if not isinstance(v,dict) else v['median'])"
timeout 300 python tools/fde_scan_probe.py 1250000 "variants 0 3 4 5" > gpurun_out/r5f_fde_scan_probe.jsonl 2>gpurun_out/r5f_probe.err; cat gpurun_out/r5f_fde_scan_probe.jsonl
MV_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --pages 200000 --steps 30 --warmup 5 --cpu-sample-pages 1024 --no-aux > gpurun_out/r5f_bench_rccl1.txt 2>gpurun_out/r5f_bench_rccl1.err
tail -c 1900 gpurun_out/r5f_bench_rccl1.txt; tail -3 gpurun_out/r5f_bench_rccl1.err
MV_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --backend gloo --pages 400000 --steps 20 --warmup 5 --cpu-sample-pages 1024 --no-aux > gpurun_out/r5f_bench_2rank_gloo.txt 2>gpurun_out/r5f_bench_2rank_gloo.err
tail -c 1900 gpurun_out/r5f_bench_2rank_gloo.txt
