#!/bin/bash
# rocprofv3 kernel trace + stats of the variant bench (per-kernel durations); summary -> gpurun_out/trace_variants.json
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/trace_v
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_v -- python $GRAFT_REPO_ROOT/tools/variant_bench.py --pages 200000 --variants 3 --rounds 5 --aux > $OUT/trace_variants.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/trace_v $OUT/trace_variants.json > /dev/null 2>&1
f=$(find /tmp/trace_v -name "*kernel_stats.csv" | head -1); cp $f $OUT/trace_variants_kernel_stats.csv; cut -c1-150 $f | head -30
