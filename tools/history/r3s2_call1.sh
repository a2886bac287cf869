#!/bin/bash
# Session-2 GPU call: the whole GPU suite on the rebuilt library, the pre-binned selection against the three-pass one at
# the full-shard shape, the store's pipelined coalescer at the plugin boundary.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(timeout 1500 python -m pytest tests -q -m gpu --timeout 1200 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^$") > $OUT/pytest_gpu_s2b.log 2>&1
tail -6 $OUT/pytest_gpu_s2b.log
timeout 600 python tools/select_fuse_probe.py 1250000 > $OUT/select_fuse_probe_1250k.json 2> $OUT/select_fuse_probe.err; tail -2 $OUT/select_fuse_probe.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/select_fuse_probe_1250k.json'))
for c in ('coarse75','coarse1000'):
    for k,v in d[c].items(): print(c, k, v)
PY
if [ "${1:-}" = "serve" ]; then
timeout 600 python tools/serve_bench.py --mode fde_then_float --pages 200000 --clients 1,8,32,128 --seconds 2 --out gpurun_out/serve_bench_fde_200k_s2b.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('direct', d.get('direct_single'), d.get('direct_batch'))
for r in d['runs']: print(r['coalescer'], r['clients'], r['requests_per_s'], r['p50_ms'], r['p99_ms'], r.get('mean_batch'))
print('lone', d.get('lone_request_overhead_over_device_ms'), 'best/direct', d.get('best_store_vs_direct_batch'))
"
fi
