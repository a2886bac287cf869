#!/bin/bash
# Full-size runs of the secondary workloads (configs[1], [3] shard shape, [4], MultiVectorStore shape).
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
for W in embed fp8 binary fde_fp8; do
  EXTRA=""
  [ "$W" = "fde_fp8" ] && EXTRA="--pages 1250000"
  (time timeout 900 python bench.py --workload $W --no-aux $EXTRA > $OUT/bench_workload_$W.json 2> $OUT/bench_workload_$W.err) 2>&1 | grep real
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_workload_$W.json"))
    r=d.get("roofline") or {}
    print("$W", d["value"], d["unit"], "ms/step", d["ms_per_step"], "roofline", r.get("achieved"), r.get("unit"), r.get("frac"), "recall", d.get("recall_at_10"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$W FAILED", e); print(open("gpurun_out/bench_workload_$W.err").read()[-1500:])
PY
done
