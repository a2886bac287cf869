#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_store.py -x -q -m gpu -k "fde_scan_ldsdma or fde_coarse_scan_and_pipeline or gpu_sharded_searcher or fde_batched_coarse_scan_matches or binary" > gpurun_out/r5g_tests.log 2>&1
echo "tests rc=$?" ; tail -4 gpurun_out/r5g_tests.log
for ru in 0 8 32; do
  env $( [ "$ru" != 0 ] && echo MV_FDE_SCAN_RU=$ru ) timeout 300 python tools/scan_ceiling_probe.py 25.6 4 >> gpurun_out/r5g_scan_ceiling.jsonl 2>gpurun_out/r5g_ceiling.err
done
cat gpurun_out/r5g_scan_ceiling.jsonl
timeout 300 python tools/fde_scan_probe.py 1250000 "variants 0 3 4 5" > gpurun_out/r5g_fde_scan_probe.jsonl 2>gpurun_out/r5g_probe.err; cat gpurun_out/r5g_fde_scan_probe.jsonl
