#!/bin/bash
# Round-3 check on the GPU box: the GPU suite, then the default bench (wall time + JSON kept under gpurun_out/).
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
free -g | head -2
(timeout 1200 python -m pytest tests -x -q -m gpu --timeout 900 2>&1 | tail -15) > $OUT/pytest_gpu.log 2>&1
tail -8 $OUT/pytest_gpu.log
( time timeout 1200 python bench.py ${BENCH_ARGS:-} > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
tail -12 $OUT/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print('bench:', d['value'], 'pages/s', d['ms_per_step'], 'ms/step', 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'recall', d['recall_at_10'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
a=d.get('aux_paths',{})
print('aux keys:', list(a.keys()))
for k in ('truth','batched_float'):
    print(k, json.dumps(a.get(k))[:600])
fs=a.get('full_shard',{})
print('full_shard:', json.dumps({k:v for k,v in fs.items() if k!='recall_at_10_vs_exact_bf16'})[:3000])
print('full_shard recall:', json.dumps(fs.get('recall_at_10_vs_exact_bf16'))[:3000])
print('two tier:', json.dumps(a.get('fp8_then_float'))[:3000])
PY
