#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_store.py -m gpu -q --timeout 600 -x -k "${1:-fde_batched}" 2>&1 | tail -30) > $OUT/pytest_quick.log 2>&1
tail -25 $OUT/pytest_quick.log
