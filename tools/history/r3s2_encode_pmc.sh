#!/bin/bash
# SQ counters (counters only, two separate passes) over the two FDE document-encode kernels -> gpurun_out/pmc_fde_encode_{1,2}.json
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmcenc_$i
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmcenc_$i -- python $R/tools/fde_encode_probe.py 20000 > $OUT/pmc_fde_encode_$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmcenc_$i $OUT/pmc_fde_encode_$i.json > /dev/null 2>&1
  tail -1 $OUT/pmc_fde_encode_$i.log | cut -c1-200
done
python - <<'PY'
import json
for i in (1, 2):
    try:
        d = json.load(open(f'gpurun_out/pmc_fde_encode_{i}.json'))
    except Exception as e:
        print(i, 'missing', e); continue
    c = d.get('counters') or {}
    for k, v in c.items():
        if 'fde_encode' in k:
            print(i, k[:70], {a: (round(b.get('avg', b) if isinstance(b, dict) else b, 1)) for a, b in v.items()} if isinstance(v, dict) else v)
PY
