#!/bin/bash
# per-kernel durations of the FDE -> rerank pipeline (rocprofv3 kernel trace of tools/fde_pipeline_probe.py)
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/tr_fde
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_fde -- python $R/tools/fde_pipeline_probe.py ${1:-200000} > $OUT/trace_fde.log 2>&1
f=$(find /tmp/tr_fde -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprofv3_kernel_stats_fde_pipeline.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f}")
PY
