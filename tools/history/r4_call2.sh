#!/bin/bash
# round 4, GPU call 2: the budget / shard-scale tests, then the default bench with the exact shard at a cautious pin fraction
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo skip-tests

( while true; do cat /sys/fs/cgroup/memory.current 2>/dev/null; sleep 2; done ) > gpurun_out/r4_memcurrent.log 2>&1 &
MON=$!
timeout 1200 python bench.py --steps 20 --warmup 5 --exact-shard-pin-frac ${PIN_FRAC:-0.7} > gpurun_out/r4_bench_a.json 2> gpurun_out/r4_bench_a.err
kill $MON
tail -c 600 gpurun_out/r4_bench_a.err
sort -n gpurun_out/r4_memcurrent.log | tail -1
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4_bench_a.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "vs_baseline", "recall_at_10")})
print(json.dumps(d["cpu_baseline"], indent=0)[:1500])
a = d.get("aux_paths", {})
print(list(a))
print(json.dumps(a.get("exact_shard", {}), indent=0)[:6000])
PY
