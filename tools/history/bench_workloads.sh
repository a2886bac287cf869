#!/bin/bash
# Full-size runs of the secondary workloads (not the driver's default bench): one JSON each in gpurun_out/
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
for W in "fp8 1000000" "binary 1000000" "fde_fp8 1250000"; do
  set -- $W
  timeout 900 python bench.py --workload $1 --pages $2 --steps 20 --warmup 3 --no-aux > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  tail -2 $OUT/bench_$1.err; cut -c1-300 $OUT/bench_$1.json; echo
done
