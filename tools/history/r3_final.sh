#!/bin/bash
# Round-3 measurement call on the FINAL library: smoke, the whole GPU suite (full log), PMC traffic passes (the record is
# hash-tied to the library and written where bench.py looks for it), the default bench, the same command under
# rocprofv3 --kernel-trace --stats, the multi-rank code paths on one GPU, and the serving benchmark of the exact-scan store.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT profiles/r3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(cd $R && timeout 1800 python -m pytest tests -q -m gpu --timeout 1500 -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^$") > $OUT/pytest_gpu_full.log 2>&1
grep -E "cosine|batched two-stage at R|hard negatives:|unplanted|passed|failed" $OUT/pytest_gpu_full.log | tail -14
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python $R/tools/variant_bench.py --pages 50000 --variants 6 --rounds 3 --no-batch > $OUT/pmc_$C.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmc_$C $OUT/rocprofv3_pmc_${C}_summary_r3.json > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $OUT/rocprofv3_pmc_FETCH_SIZE_summary_r3.json $OUT/rocprofv3_pmc_WRITE_SIZE_summary_r3.json 50000 $OUT/pmc_traffic_r3.json && cp $OUT/pmc_traffic_r3.json $R/profiles/r3/pmc_traffic_r3.json
cd $R
( time timeout 1200 python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench_1gpu.err ) 2>&1 | grep real
tail -3 $OUT/bench_1gpu.err; cut -c1-400 $OUT/bench_1gpu.json
(MV_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --backend gloo --pages 200000 --steps 10 --warmup 2 --no-aux > $OUT/bench_2rank_selfspawn.json 2> $OUT/bench_2rank.err); echo "2-rank rc=$?"
(MV_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --pages 200000 --steps 10 --warmup 2 --no-aux 2>$OUT/bench_rccl1.err | grep '^{' > $OUT/bench_rccl1.json)
python - <<'PY'
import json
for f in ('bench_2rank_selfspawn', 'bench_rccl1'):
    try:
        d=json.load(open(f'gpurun_out/{f}.json'))
        print(f, d['n_gpus'], d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'], 'recall', d['recall_at_10'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'], 'err', d['max_rel_score_err_vs_oracle'], d['config'].get('collective_and_merge_ms_per_step'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
cd /tmp
rm -rf /tmp/tr_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bench -- python $R/bench.py --no-aux --no-cpu-baseline > $OUT/bench_1gpu_under_rocprof.json 2> $OUT/rocprof_bench.err
python $R/tools/rocprof_summary.py /tmp/tr_bench $OUT/rocprofv3_kernel_trace_summary_bench_1M_r3.json > /dev/null 2>&1
f=$(find /tmp/tr_bench -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprofv3_kernel_stats_bench_1M_r3.csv; cut -c1-160 $f | head -5
cd $R
(timeout 600 python tools/serve_bench.py --mode float --pages 1000000 --clients 1,8,32 --seconds 2 --out gpurun_out/serve_bench_float_1M.json 2>/dev/null | cut -c1-600)
