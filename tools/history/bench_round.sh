#!/bin/bash
# Headline bench (1 GPU, BASELINE configs[2]) + rocprofv3 kernel stats of the same command + PMC HBM traffic passes.
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout 900 python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench_1gpu.err); tail -3 $OUT/bench_1gpu.err; cut -c1-400 $OUT/bench_1gpu.json
cd /tmp
rm -rf /tmp/tr_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bench -- python $R/bench.py > $OUT/bench_1gpu_under_rocprof.json 2> $OUT/rocprof_bench.err
python $R/tools/rocprof_summary.py /tmp/tr_bench $OUT/rocprofv3_kernel_trace_summary_bench_1M.json > /dev/null 2>&1
f=$(find /tmp/tr_bench -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprofv3_kernel_stats_bench_1M.csv; cut -c1-160 $f | head -8
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python $R/tools/variant_bench.py --pages 50000 --variants 6 --rounds 3 > $OUT/pmc_$C.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmc_$C $OUT/rocprofv3_pmc_${C}_summary.json > /dev/null 2>&1
done
