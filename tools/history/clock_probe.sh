#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/ck1 /tmp/ck2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ck1 -- python $GRAFT_REPO_ROOT/tools/clock_probe.py > $OUT/ck1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/ck2 -- python $GRAFT_REPO_ROOT/tools/clock_probe.py > $OUT/ck2.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/ck1 $OUT/ck1.json > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/ck2 $OUT/ck2.json > /dev/null 2>&1
python - <<'PY'
import json
t=json.load(open('gpurun_out/ck1.json'))['kernel_trace_avg_us']; c=json.load(open('gpurun_out/ck2.json'))['counters']
for k in t:
    if k in c and any(x in k for x in ('batch_kernel','mfma_peak','ldsdma')):
        g=c[k]['GRBM_GUI_ACTIVE']['avg']; us=t[k]['avg_us']
        print(k[:60], 'avg_us', round(us,1), 'GUI_ACTIVE', round(g), 'cycles/us', round(g/us,1), '-> per-XCD GHz if summed over 8:', round(g/us/8/1000,3), ' raw GHz:', round(g/us/1000,3))
PY
