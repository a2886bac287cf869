#!/bin/bash
# batched FDE pipeline: parity tests, probe, kernel trace of the probe
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "fde_batched or large_k_radix or topk_ties or fde_coarse_scan or batched_queries_on_a_uniform" 2>&1 | tail -15) > $OUT/pytest_fde_batch.log 2>&1
tail -6 $OUT/pytest_fde_batch.log
timeout 600 python tools/fde_batch_probe.py ${1:-200000} > $OUT/fde_batch_probe.json 2> $OUT/fde_batch_probe.err || tail -5 $OUT/fde_batch_probe.err
cat $OUT/fde_batch_probe.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fde_batch -- python $OLDPWD/tools/fde_batch_probe.py ${1:-200000} > /dev/null 2> $OUT/prof_fde_batch.err
cd $OLDPWD
f=$(find /tmp/prof_fde_batch -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" $OUT/rocprofv3_kernel_stats_fde_batch_probe.csv; head -14 "$f" | cut -c1-200; fi
