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
