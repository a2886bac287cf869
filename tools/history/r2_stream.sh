#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --timeout 600 -k "persistent_stream or all_variants_small" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40) > $OUT/pytest_stream.log 2>&1
tail -4 $OUT/pytest_stream.log
python tools/variant_bench.py --pages ${1:-400000} --variants 6,14 --rounds 8 --no-batch 2>&1 | grep "^q="
