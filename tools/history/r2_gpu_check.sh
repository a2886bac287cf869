#!/bin/bash
# Round-2 GPU check: new sharded / pad-rule / concurrency tests first (fail fast), then the whole GPU suite.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q --timeout 600 2>&1 | tail -40) > $OUT/pytest_sharded.log 2>&1
tail -25 $OUT/pytest_sharded.log
(timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -60) > $OUT/pytest_gpu.log 2>&1
tail -30 $OUT/pytest_gpu.log
