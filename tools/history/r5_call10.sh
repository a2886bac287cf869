#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exact_tier.py tests/test_gpu_store.py -x -q -m gpu -k "fde_scan_row_quarters or fde_coarse_scan_and_pipeline or caller_supplied_fde or fde_batched or gpu_sharded_searcher or ingest_runs_beside or empty or filter" > gpurun_out/r5j_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5j_tests.log
for st in 1 0; do
timeout 300 python tools/fde_chain_probe.py 200000 60 $st >> gpurun_out/r5j_fde_chain.jsonl 2>gpurun_out/r5j_chain.err
timeout 300 python tools/fde_chain_probe.py 1250000 40 $st >> gpurun_out/r5j_fde_chain.jsonl 2>>gpurun_out/r5j_chain.err
done
cat gpurun_out/r5j_fde_chain.jsonl
for st in 1 0; do
cd /tmp; rm -rf /tmp/chain_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/chain_trace -- python $R/tools/fde_chain_probe.py 200000 30 $st > $R/gpurun_out/r5j_chain_trace_$st.log 2>&1
cd $R
python tools/fde_chain_trace_summary.py /tmp/chain_trace gpurun_out/r5j_rocprofv3_kernel_trace_fde_request_chain_200k_stats$st.json > /dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r5j_rocprofv3_kernel_trace_fde_request_chain_200k_stats$st.json"))
print("stats=$st", {k:v for k,v in d.items() if k!="chain"})
for c in d.get("chain",[]): print("   ", c)
PY
done
