#!/bin/bash
# functional checks of bench.py paths on the 1-GPU box: aux paths at small size; 2 ranks sharing the GPU over gloo
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python bench.py --pages 200000 --steps 10 --aux-pages 100000 > $OUT/bench_small.json 2> $OUT/bench_small.err; tail -2 $OUT/bench_small.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_small.json'))
print(d['value'], d['roofline']['achieved'], d['recall_at_10'])
print(json.dumps(d.get('aux_paths'),indent=0)[:3000])
PY
MV_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --pages 200000 --backend gloo > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; tail -4 $OUT/bench_2rank_gloo.err; cut -c1-600 $OUT/bench_2rank_gloo.json
