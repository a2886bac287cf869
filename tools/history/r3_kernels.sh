#!/bin/bash
# Round-3 kernel iteration: parity of the new forms (batched fp8 scan, 32-page FDE batch tiles), their sweeps, and a kernel
# trace of the batched selection chain.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --timeout 800 -k "batched_fp8 or fde_batch or fde_scan or batched_pipeline or two_tier or fp8_then_float" 2>&1 | tail -6) | tee $OUT/pytest_kernels.log
for N in 200000 1250000; do
  for F in default half_tiles; do
    (cd $R && timeout 300 python tools/fde_batch_scan_probe.py $N $F 2>/dev/null | tail -1) | tee -a $OUT/fde_batch_scan_forms_r3.jsonl
  done
done
(cd $R && timeout 600 python tools/variant_bench.py --pages 200000 --variants 6 --rounds 4 --batch-only --out gpurun_out/variants_batch_r3b.json 2>&1 | grep "^batch\|^mfma") | tee $OUT/variants_batch_b.log
cd /tmp && rm -rf /tmp/seltrace && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/seltrace -- python $R/tools/select_trace_probe.py 1250000 > $OUT/select_trace_probe.log 2>&1
tail -1 $OUT/select_trace_probe.log
python $R/tools/rocprof_summary.py /tmp/seltrace $OUT/rocprofv3_kernel_trace_select_probe_r3.json > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/rocprofv3_kernel_trace_select_probe_r3.json'))
for k,v in list(d['kernel_trace_avg_us'].items())[:14]:
    print(f"{v['avg_us']:10.1f} us x {v['calls']:5d}  {k[:110]}")
PY
