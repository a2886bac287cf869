#!/bin/bash
# Round-2 measurement call: default bench, the same command under rocprofv3 kernel-trace, PMC traffic passes, 2-rank self-spawn check.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && time timeout 1200 python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench_1gpu.err); tail -4 $OUT/bench_1gpu.err; cut -c1-600 $OUT/bench_1gpu.json
# plain N>1 invocation: self-spawns 2 ranks sharing the one GPU (gloo): functional check of the launcher path
(cd $R && MV_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --backend gloo --pages 200000 --steps 10 --warmup 2 > $OUT/bench_2rank_selfspawn.json 2> $OUT/bench_2rank.err); echo "2-rank rc=$?"; cut -c1-300 $OUT/bench_2rank_selfspawn.json
(cd $R && MV_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --backend gloo --workload fde_fp8 --pages 200000 --steps 10 --warmup 2 > $OUT/bench_2rank_fde_selfspawn.json 2>> $OUT/bench_2rank.err); echo "2-rank fde rc=$?"
# 1-rank RCCL communicator (nccl backend path: GpuShardedSearcher / GpuTwoStageSearcher)
(cd $R && MV_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --pages 200000 --steps 10 --warmup 2 --no-aux --no-cpu-baseline 2>$OUT/bench_rccl1.err | grep '^{' > $OUT/bench_rccl1.json)
(cd $R && MV_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --workload fde_fp8 --pages 200000 --steps 10 --warmup 2 --no-aux 2>$OUT/bench_rccl1_fde.err | grep '^{' > $OUT/bench_rccl1_fde.json)
python - <<'PY'
import json
for f in ('bench_rccl1', 'bench_rccl1_fde', 'bench_2rank_fde_selfspawn'):
    try:
        d=json.load(open(f'gpurun_out/{f}.json'))
        print(f, d['n_gpus'], d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'], 'recall', d['recall_at_10'], d['config'].get('rccl_ranks'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
cd /tmp
rm -rf /tmp/tr_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bench -- python $R/bench.py --no-aux --no-cpu-baseline > $OUT/bench_1gpu_under_rocprof.json 2> $OUT/rocprof_bench.err
python $R/tools/rocprof_summary.py /tmp/tr_bench $OUT/rocprofv3_kernel_trace_summary_bench_1M.json > /dev/null 2>&1
f=$(find /tmp/tr_bench -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprofv3_kernel_stats_bench_1M.csv; cut -c1-160 $f | head -6
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python $R/tools/variant_bench.py --pages 50000 --variants 6 --rounds 3 --no-batch > $OUT/pmc_$C.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmc_$C $OUT/rocprofv3_pmc_${C}_summary.json > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $OUT/rocprofv3_pmc_FETCH_SIZE_summary.json $OUT/rocprofv3_pmc_WRITE_SIZE_summary.json 50000 $OUT/pmc_traffic_r2.json
