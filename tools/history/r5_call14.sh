#!/bin/bash
# round 5, call 14: the e4m3 scan with two consecutive pages per workgroup -- parity, then A/B against one page per workgroup.
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exact_tier.py -x -q -m gpu -k "fp8" 2>&1 | tail -4 > gpurun_out/r5n_fp8_tests.log
echo "fp8 tests rc=$?"; tail -2 gpurun_out/r5n_fp8_tests.log
for rep in 1 2; do
  for pairs in 0 1; do
    MV_FP8_SCAN_PAIRS=$pairs timeout 200 python bench.py --workload fp8 --pages 1250000 --steps 30 --warmup 3 --no-aux --no-cpu-baseline 2>gpurun_out/r5n_bench.err | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'pairs': $pairs, 'rep': $rep, 'ms_per_step': d['ms_per_step'], 'roofline': d['roofline']}))" | tee -a gpurun_out/r5n_fp8_pairs_ab.jsonl
  done
done
