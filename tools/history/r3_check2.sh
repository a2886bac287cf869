#!/bin/bash
# GPU suite + sanitizer runs + default bench.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -x -q -m gpu --timeout 1200 -s 2>&1 | grep -v "^$" | tail -40) > $OUT/pytest_gpu.log 2>&1
grep -E "cosine|batched two-stage|passed|failed|Error|error" $OUT/pytest_gpu.log | tail -20
bash tools/r3_sanitize.sh 2>&1 | tail -16
( time timeout 1200 python bench.py ${BENCH_ARGS:-} > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
tail -6 $OUT/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print('bench:', d['value'], 'pages/s', d['ms_per_step'], 'ms/step', 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'recall', d['recall_at_10'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
a=d.get('aux_paths',{})
print('aux keys:', list(a.keys()))
fs=a.get('full_shard',{})
print('full_shard:', json.dumps({k:v for k,v in fs.items() if k not in ('recall_at_10_vs_exact_bf16','fde_then_fp8_rerank','note')})[:2500])
r=fs.get('recall_at_10_vs_exact_bf16',{})
print('full_shard recall:', json.dumps({k:v for k,v in r.items() if k!='by_margin_hard_and_clustered'})[:2500])
print('serving:', json.dumps(a.get('serving'))[:3500])
PY
