#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
(timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -x -k "${1:-sharded_store_answers}" -s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -30) > $OUT/pytest_quick.log 2>&1
tail -12 $OUT/pytest_quick.log
