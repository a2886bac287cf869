#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q --timeout 600 -k "topk or radix or fde or two_stage or shard_comm or candidates" 2>&1 | tail -8) > $OUT/pytest_fde.log 2>&1
tail -4 $OUT/pytest_fde.log
timeout 600 python tools/fde_pipeline_probe.py 200000 2>&1 | tail -1 | tee $OUT/fde_pipeline_probe_200k.json
timeout 600 python tools/fde_pipeline_probe.py 1000000 2>&1 | tail -1 | tee $OUT/fde_pipeline_probe_1M.json
