#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
t0=$(date +%s)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r5t_pytest_gpu_full.log 2>&1
echo "gpu suite rc=$? wall=$(( $(date +%s) - t0 )) s"
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^$" gpurun_out/r5t_pytest_gpu_full.log | tail -16
