#!/bin/bash
# Round-4 measurement call on the FINAL library: smoke, PMC traffic passes (the record is hash-tied to the library and written where
# bench.py looks for it), the default bench (what the driver runs), the same command under rocprofv3 --kernel-trace --stats, SQ counters
# of the sign-bit scan and of the batched FDE coarse pass (counter-only passes), the 2-rank and the RCCL 1-rank code paths.
#   SUITE=1 also runs the whole GPU suite first.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT profiles/r4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
if [ "${SUITE:-0}" = "1" ]; then
  (cd $R && timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^$") > $OUT/r4_pytest_gpu_full.log 2>&1
  tail -3 $OUT/r4_pytest_gpu_full.log
fi
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python $R/tools/variant_bench.py --pages 50000 --variants 6 --rounds 3 --no-batch > $OUT/pmc_$C.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmc_$C $OUT/rocprofv3_pmc_${C}_summary_r4.json > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $OUT/rocprofv3_pmc_FETCH_SIZE_summary_r4.json $OUT/rocprofv3_pmc_WRITE_SIZE_summary_r4.json 50000 $OUT/pmc_traffic_r4.json && cp $OUT/pmc_traffic_r4.json $R/profiles/r4/pmc_traffic_r4.json
cd $R
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/r4_bench_1gpu.json 2> $OUT/r4_bench_1gpu.err ) 2>&1 | grep real
tail -2 $OUT/r4_bench_1gpu.err | cut -c1-300; cut -c1-500 $OUT/r4_bench_1gpu.json
cd /tmp
rm -rf /tmp/tr_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bench -- python $R/bench.py --steps 20 --warmup 5 --no-aux --no-cpu-baseline > $OUT/r4_bench_1gpu_under_rocprof.json 2> $OUT/r4_rocprof_bench.err
python $R/tools/rocprof_summary.py /tmp/tr_bench $OUT/rocprofv3_kernel_trace_summary_bench_1M_r4.json > /dev/null 2>&1
f=$(find /tmp/tr_bench -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprofv3_kernel_stats_bench_1M_r4.csv; cut -c1-200 $f | head -4
# SQ counters (counters only, separate passes): the sign-bit scan (1 M pages, default variant) and the batched FDE coarse pass (1.25 M pages, 32 requests)
i=0
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmcbin_$i /tmp/pmcfde_$i
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmcbin_$i -- python $R/tools/binary_probe.py 1000000 4 > $OUT/r4_pmc_binary_$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmcbin_$i $OUT/r4_pmc_binary_$i.json > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmcfde_$i -- python $R/tools/fde_batch_scan_probe.py 1250000 > $OUT/r4_pmc_fde_batch_$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmcfde_$i $OUT/r4_pmc_fde_batch_$i.json > /dev/null 2>&1
done
cd $R
(MV_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --backend gloo --pages 200000 --steps 10 --warmup 2 --no-aux > $OUT/r4_bench_2rank_selfspawn.json 2> $OUT/r4_bench_2rank.err); echo "2-rank rc=$?"
(MV_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --pages 200000 --steps 10 --warmup 2 --no-aux 2>$OUT/r4_bench_rccl1.err | grep '^{' > $OUT/r4_bench_rccl1.json)
python - <<'PY'
import json
for f in ('r4_bench_2rank_selfspawn', 'r4_bench_rccl1'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][-1])
        print(f, d['n_gpus'], d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'], 'recall', d['recall_at_10'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'], 'err', d['max_rel_score_err_vs_oracle'], d['config'].get('collective_and_merge_ms_per_step'))
    except Exception as e:
        print(f, 'FAILED', e)
for name in ('binary', 'fde_batch'):
    for i in (1, 2):
        try:
            c = json.load(open(f'gpurun_out/r4_pmc_{name}_{i}.json'))['counters']
        except Exception as e:
            print(name, i, 'missing', e); continue
        for k, v in c.items():
            if ('binary' in k and 'maxsim' in k) or 'fde_scan_batch' in k:
                print(name, i, k[:70], {a: round(b['avg'], 1) for a, b in v.items()})
PY
