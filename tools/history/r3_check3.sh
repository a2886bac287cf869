#!/bin/bash
# GPU suite + default bench (no sanitizer runs: TSan/ASan against the real ROCm runtime took a node down).
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -x -q -m gpu --timeout 1200 -s 2>&1 | grep -v "^$" | tail -40) > $OUT/pytest_gpu.log 2>&1
grep -E "cosine|batched two-stage|passed|failed|Error|error" $OUT/pytest_gpu.log | tail -20
( time timeout 1200 python bench.py ${BENCH_ARGS:-} > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
tail -6 $OUT/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print('bench:', d['value'], 'pages/s', d['ms_per_step'], 'ms/step', 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'recall', d['recall_at_10'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
a=d.get('aux_paths',{})
print('aux keys:', list(a.keys()))
fs=a.get('full_shard',{})
print('full_shard:', json.dumps({k:v for k,v in fs.items() if k not in ('recall_at_10_vs_exact_bf16','note')})[:3500])
print('fde enc:', json.dumps(a.get('fde_document_encode')))
print('serving:', json.dumps(a.get('serving'))[:3500])
PY
