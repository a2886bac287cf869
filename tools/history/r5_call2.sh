#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 500 python tools/r5_call2.py > gpurun_out/r5b_scan_ceiling.json 2> gpurun_out/r5b_scan_ceiling.err
cat gpurun_out/r5b_scan_ceiling.err | tail -8
: > gpurun_out/r5b_fde_scan_probe.jsonl
for n in 2500000 5000000; do
  timeout 400 python tools/fde_scan_probe.py $n "pages=$n" >> gpurun_out/r5b_fde_scan_probe.jsonl 2>>gpurun_out/r5b_probe.err
done
MV_FDE_SCAN_BLOCKS_PER_CU=1 timeout 400 python tools/fde_scan_probe.py 5000000 "pages=5000000 bpc=1" >> gpurun_out/r5b_fde_scan_probe.jsonl 2>>gpurun_out/r5b_probe.err
cat gpurun_out/r5b_fde_scan_probe.jsonl
