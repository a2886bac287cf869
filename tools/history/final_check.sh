#!/bin/bash
# Round-end dress rehearsal on the GPU box: smoke, GPU tests, default bench, 2-rank functional run.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3)
(time python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_final.json'))
print('bench:', d['value'], 'pages/s', d['ms_per_step'], 'ms/step', 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'recall', d['recall_at_10'], 'cpu', d['cpu_baseline']['value'])
print('aux:', {k:(v.get('pages_per_s') or v) for k,v in d['aux_paths'].items() if isinstance(v,dict) and k!='batched_float'}, d['aux_paths']['batched_float']['B16']['TFLOPs'])
PY
MV_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --pages 200000 --backend gloo 2>$OUT/bench_2rank.err | grep '^{' > $OUT/bench_2rank.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_2rank.json'))
print('2-rank gloo functional:', d['n_gpus'], d['value'], 'recall', d['recall_at_10'], d['config']['parallelism'])
PY
# 1-rank RCCL communicator: the nccl-backend code path (GpuShardedSearcher: collectives + mv_merge_topk; two-stage FDE pipeline)
MV_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --pages 200000 --steps 10 --warmup 2 --no-aux 2>$OUT/bench_rccl1.err | grep '^{' > $OUT/bench_rccl1.json
MV_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --workload fde_fp8 --pages 200000 --steps 10 --warmup 2 --no-aux 2>$OUT/bench_rccl1_fde.err | grep '^{' > $OUT/bench_rccl1_fde.json
python - <<'PY'
import json
for f in ('bench_rccl1', 'bench_rccl1_fde'):
    d=json.load(open(f'gpurun_out/{f}.json'))
    print(f, 'RCCL 1-rank:', d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'], 'recall', d['recall_at_10'])
PY
