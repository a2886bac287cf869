#!/bin/bash
# Round-5 re-measurement after the e4m3 scan moved to page pairs (the library changed, so the hash-tied PMC record is re-taken):
# smoke, PMC traffic passes of the headline kernel, PMC traffic of the secondary scans, the default bench, the same under rocprofv3
# --kernel-trace --stats, then the whole GPU suite.  (SQ counters of the sign-bit / batched FDE kernels and the 2-rank / 1-rank-RCCL lines
# are tools/r5_final.sh's: those kernels and paths did not change.)
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT profiles/r5
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python $R/tools/variant_bench.py --pages 50000 --variants 6 --rounds 3 --no-batch > $OUT/r5_pmc_$C.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmc_$C $OUT/rocprofv3_pmc_${C}_summary_r5.json > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $OUT/rocprofv3_pmc_FETCH_SIZE_summary_r5.json $OUT/rocprofv3_pmc_WRITE_SIZE_summary_r5.json 50000 $OUT/pmc_traffic_r5.json && cp $OUT/pmc_traffic_r5.json $R/profiles/r5/pmc_traffic_r5.json
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ax_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/ax_$C -- python $R/tools/r4_aux_traffic_probe.py 400000 > $OUT/r5_ax_$C.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/ax_$C $OUT/r5_aux_pmc_$C.json > /dev/null 2>&1
done
python $R/tools/r4_aux_traffic.py $OUT/r5_aux_pmc_FETCH_SIZE.json $OUT/r5_aux_pmc_WRITE_SIZE.json 400000 $OUT/pmc_traffic_aux_scans_r5.json | head -30
cd $R
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r5_bench_1gpu_stdout.txt 2> $OUT/r5_bench_1gpu_stderr.txt
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s; stdout lines (bytes): $(awk '{printf "%d ", length($0)}' $OUT/r5_bench_1gpu_stdout.txt); JSON-shaped stderr lines: $(grep -c '^{' $OUT/r5_bench_1gpu_stderr.txt)"
tail -n 1 $OUT/r5_bench_1gpu_stdout.txt
cd /tmp
rm -rf /tmp/tr_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bench -- python $R/bench.py --steps 20 --warmup 5 --no-aux --no-cpu-baseline > $OUT/r5_bench_1gpu_under_rocprof.txt 2> $OUT/r5_rocprof_bench.err
python $R/tools/rocprof_summary.py /tmp/tr_bench $OUT/rocprofv3_kernel_trace_summary_bench_1M_r5.json > /dev/null 2>&1
f=$(find /tmp/tr_bench -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprofv3_kernel_stats_bench_1M_r5.csv; cut -c1-220 $f | head -4
tail -n 1 $OUT/r5_bench_1gpu_under_rocprof.txt | cut -c1-400
cd $R
t0=$(date +%s)
timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r5u_pytest_gpu_full.log 2>&1
echo "gpu suite rc=$? wall=$(( $(date +%s) - t0 )) s"
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^$" gpurun_out/r5u_pytest_gpu_full.log | tail -14
