#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r5k_pytest_gpu_full.log 2>&1
echo "gpu suite rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -22 gpurun_out/r5k_pytest_gpu_full.log
for st in 1 0; do
timeout 300 python tools/fde_chain_probe.py 200000 60 $st >> gpurun_out/r5k_fde_chain.jsonl 2>gpurun_out/r5k_chain.err
done
cat gpurun_out/r5k_fde_chain.jsonl
cd /tmp; rm -rf /tmp/chain_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/chain_trace -- python $R/tools/fde_chain_probe.py 200000 30 0 > $R/gpurun_out/r5k_chain_trace_0.log 2>&1
cd $R
python tools/fde_chain_trace_summary.py /tmp/chain_trace gpurun_out/r5k_rocprofv3_kernel_trace_fde_request_chain_200k_stats0.json > /dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r5k_rocprofv3_kernel_trace_fde_request_chain_200k_stats0.json"))
print({k:v for k,v in d.items() if k!="chain"})
for c in d.get("chain",[]): print("   ", c)
PY
