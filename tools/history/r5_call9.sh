#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exact_tier.py -x -q -m gpu -k "fde_scan_row_quarters or fde_coarse_scan_and_pipeline or caller_supplied_fde or fde_batched" > gpurun_out/r5i_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5i_tests.log
timeout 300 python tools/fde_chain_probe.py 200000 60 > gpurun_out/r5i_fde_chain_200k.json 2>gpurun_out/r5i_chain.err; cat gpurun_out/r5i_fde_chain_200k.json
timeout 300 python tools/fde_chain_probe.py 1250000 40 > gpurun_out/r5i_fde_chain_1250k.json 2>>gpurun_out/r5i_chain.err; cat gpurun_out/r5i_fde_chain_1250k.json
cd /tmp; rm -rf /tmp/chain_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/chain_trace -- python $R/tools/fde_chain_probe.py 200000 30 > $R/gpurun_out/r5i_chain_trace.log 2>&1
cd $R
python tools/fde_chain_trace_summary.py /tmp/chain_trace gpurun_out/r5i_rocprofv3_kernel_trace_fde_request_chain_200k.json | head -120
