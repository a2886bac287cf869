#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "fde_batched" 2>&1 | tail -5) > $OUT/pytest_fde_batch.log 2>&1
tail -3 $OUT/pytest_fde_batch.log
for n in 200000 1000000; do
  timeout 300 python tools/fde_batch_scan_probe.py $n 2>&1 | tail -1
  timeout 300 python tools/fde_batch_scan_probe.py $n single_tile 2>&1 | tail -1
  timeout 300 python tools/fde_batch_scan_probe.py $n hi_only 2>&1 | tail -1
done | tee $OUT/fde_scan_exp.jsonl
