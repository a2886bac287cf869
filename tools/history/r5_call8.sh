#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --durations=40 > gpurun_out/r5h_pytest_gpu_full.log 2>&1
echo "gpu suite rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -60 gpurun_out/r5h_pytest_gpu_full.log
