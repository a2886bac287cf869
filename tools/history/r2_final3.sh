#!/bin/bash
# Round-end measurement on the FINAL library: dress rehearsal + PMC passes (tools/r2_final.sh), then the default bench
# again with the fresh PMC record in place (so `roofline.traffic` is the record of THIS build), then the configs[3] workload.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/r2_final.sh
cp $OUT/pmc_traffic_r2.json profiles/r2/pmc_traffic_r2g.json
(timeout 900 python bench.py > $OUT/bench_1gpu_final.json 2> $OUT/bench_1gpu_final.err); tail -2 $OUT/bench_1gpu_final.err; cut -c1-200 $OUT/bench_1gpu_final.json
(timeout 900 python bench.py --workload fde_fp8 > $OUT/bench_fde_fp8.json 2> $OUT/bench_fde_fp8.err); tail -2 $OUT/bench_fde_fp8.err; cut -c1-200 $OUT/bench_fde_fp8.json
bash tools/r2_fde_batch.sh 200000 > $OUT/fde_batch_probe.log 2>&1; tail -3 $OUT/fde_batch_probe.log | cut -c1-300
