#!/bin/bash
# PMC passes over the final aux kernel set: HBM traffic (FETCH_SIZE) and SQ activity; summaries -> gpurun_out/pmc2_*.json
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/variant_bench.py --pages 100000 --variants 6 --rounds 2 --aux"
i=0
for SET in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc2_$i
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc2_$i -- $CMD > $OUT/pmc2_$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/pmc2_$i $OUT/pmc2_$i.json > /dev/null 2>&1
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc2_1.json'))['counters'] if False else None
PY
