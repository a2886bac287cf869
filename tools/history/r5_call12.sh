#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_store.py -x -q -m gpu -k "binary or sign or store_scenarios or hamming" > gpurun_out/r5m_binary_tests.log 2>&1
echo "binary tests rc=$?"; tail -4 gpurun_out/r5m_binary_tests.log
timeout 300 python tools/binary_probe.py 1250000 4,7 > gpurun_out/r5m_binary_probe_1250k.log 2>&1; tail -3 gpurun_out/r5m_binary_probe_1250k.log
timeout 300 python tools/binary_probe.py 1000000 4,7 64 > gpurun_out/r5m_binary_probe_1M_q64.log 2>&1; tail -3 gpurun_out/r5m_binary_probe_1M_q64.log
timeout 300 python tools/binary_probe.py 200000 4,7 > gpurun_out/r5m_binary_probe_200k.log 2>&1; tail -3 gpurun_out/r5m_binary_probe_200k.log
