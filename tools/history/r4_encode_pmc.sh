#!/bin/bash
# round 4: kernel trace + SQ counters (counters only, two separate passes) over the FDE document-encode kernels
#   -> gpurun_out/r4_fde_encode_trace.json, gpurun_out/r4_pmc_fde_encode_{1,2}.json
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/enctrace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/enctrace -- python $R/tools/fde_encode_probe.py 20000 > $OUT/r4_fde_encode_trace.log 2>&1
python $R/tools/rocprof_summary.py /tmp/enctrace $OUT/r4_fde_encode_trace.json > /dev/null 2>&1
i=0
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmcenc_$i
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmcenc_$i -- python $R/tools/fde_encode_probe.py 20000 > $OUT/r4_pmc_fde_encode_$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmcenc_$i $OUT/r4_pmc_fde_encode_$i.json > /dev/null 2>&1
done
cd $R
python - <<'PY'
import json
t = json.load(open('gpurun_out/r4_fde_encode_trace.json'))['kernel_trace_avg_us']
for k, v in t.items():
    if 'fde_' in k: print(k[:80], v)
for i in (1, 2):
    d = json.load(open(f'gpurun_out/r4_pmc_fde_encode_{i}.json')).get('counters') or {}
    for k, v in d.items():
        if 'fde_hash' in k or 'fde_project' in k or 'fde_encode_doc' in k:
            print(i, k[:60], {a: round(b['avg'], 1) for a, b in v.items()})
PY
