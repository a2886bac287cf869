#!/bin/bash
# round 5, call 1: the new FDE scan (parity + tuning sweep), the compact bench line (test + the driver's own command)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fde_scan_ldsdma or fde_coarse_scan_and_pipeline or fde_batched_coarse_scan_matches" > gpurun_out/r5a_fde_scan_tests.log 2>&1
echo "fde tests rc=$?" ; tail -3 gpurun_out/r5a_fde_scan_tests.log
: > gpurun_out/r5a_fde_scan_probe.jsonl
for cfg in "0 0" "16 0" "32 0" "64 0" "0 1" "64 1"; do
  set -- $cfg
  env $( [ "$1" != 0 ] && echo MV_FDE_SCAN_PPW=$1 ) $( [ "$2" != 0 ] && echo MV_FDE_SCAN_BLOCKS_PER_CU=$2 ) timeout 300 python tools/fde_scan_probe.py 1250000 "ppw=$1 bpc=$2" >> gpurun_out/r5a_fde_scan_probe.jsonl 2>gpurun_out/r5a_probe.err
done
cat gpurun_out/r5a_fde_scan_probe.jsonl
timeout 900 python -m pytest tests/test_bench_line.py -x -q -m gpu > gpurun_out/r5a_bench_line_test.log 2>&1
echo "bench line test rc=$?"; tail -5 gpurun_out/r5a_bench_line_test.log
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5a_bench_stdout.txt 2> gpurun_out/r5a_bench_stderr.txt
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -c 3200 gpurun_out/r5a_bench_stdout.txt
wc -c gpurun_out/r5a_bench_stdout.txt; awk '{print length($0)}' gpurun_out/r5a_bench_stdout.txt
grep -c '^{' gpurun_out/r5a_bench_stderr.txt
