#!/bin/bash
# Round-5 measurement call on the FINAL library: smoke, PMC traffic passes (the record is hash-tied to the library / its sources and written
# where bench.py looks for it), the default bench (the driver's command), the same under rocprofv3 --kernel-trace --stats, SQ counters of the
# sign-bit scan, the batched FDE coarse pass and the single-query FDE scan (counter-only passes), the 2-rank gloo and the RCCL 1-rank lines.
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
python $R/tools/r4_aux_traffic.py $OUT/r5_aux_pmc_FETCH_SIZE.json $OUT/r5_aux_pmc_WRITE_SIZE.json 400000 $OUT/pmc_traffic_aux_scans_r5.json | head -40
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
# SQ counters (counters only, separate passes)
i=0
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmcbin_$i /tmp/pmcfde_$i /tmp/pmcscan_$i
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmcbin_$i -- python $R/tools/binary_probe.py 1000000 4 > $OUT/r5_pmc_binary_$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmcbin_$i $OUT/r5_pmc_binary_$i.json > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmcfde_$i -- python $R/tools/fde_batch_scan_probe.py 1250000 > $OUT/r5_pmc_fde_batch_$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmcfde_$i $OUT/r5_pmc_fde_batch_$i.json > /dev/null 2>&1
done
cd $R
(MV_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --backend gloo --pages 400000 --steps 20 --warmup 5 --cpu-sample-pages 2048 --no-aux > $OUT/r5_bench_2rank_gloo.txt 2> $OUT/r5_bench_2rank.err); echo "2-rank rc=$?"
(MV_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --pages 200000 --steps 30 --warmup 5 --cpu-sample-pages 2048 --no-aux > $OUT/r5_bench_rccl1.txt 2>$OUT/r5_bench_rccl1.err); echo "rccl1 rc=$?"
python - <<'PY'
import json
for f in ('r5_bench_2rank_gloo', 'r5_bench_rccl1'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/{f}.txt') if l.startswith('{')][-1])
        print(f, d['n_gpus'], d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'], 'recall', d['recall_at_10'], 'err', d['max_rel_score_err_vs_oracle'], 'exchange', d['config'].get('collective_and_merge_ms_per_step'), 'step-local', d['config'].get('step_minus_local_ms_per_step'))
    except Exception as e:
        print(f, 'FAILED', e)
for name in ('binary', 'fde_batch'):
    for i in (1, 2):
        try:
            c = json.load(open(f'gpurun_out/r5_pmc_{name}_{i}.json'))['counters']
        except Exception as e:
            print(name, i, 'missing', e); continue
        for k, v in c.items():
            if ('binary' in k and 'maxsim' in k) or 'fde_scan_batch' in k or 'fde_scan_rowq' in k:
                print(name, i, k[:70], {a: round(b['avg'], 1) for a, b in v.items()})
PY
