#!/bin/bash
# round 5, call 15: rocprofv3 kernel stats of the e4m3 workload on the final library (the page-pair scan's average launch beside bench.py's HIP events).
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; rm -rf /tmp/tr_fp8
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_fp8 -- python $R/bench.py --workload fp8 --pages 1250000 --steps 20 --warmup 5 --no-aux --no-cpu-baseline > $OUT/r5o_bench_fp8_under_rocprof.txt 2> $OUT/r5o_rocprof.err
f=$(find /tmp/tr_fp8 -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprofv3_kernel_stats_bench_fp8_1250k_r5o.csv; cut -c1-200 $f | head -4
tail -n 1 $OUT/r5o_bench_fp8_under_rocprof.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
